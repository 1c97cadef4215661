// coloc_hip_types.hpp -- the types HIPDetector / HIPMatcher exchange with CoLoC.
//
// Inside the reference tree (OpenMVG present) define COLOC_HIP_WITH_OPENMVG before including the
// policy headers: the real openMVG / coloc types from include/coloc/colocData.hpp are used and this
// file adds nothing.  Stand-alone (this repository: OpenMVG is an empty submodule in the reference
// snapshot and is not installed here) the minimal stand-ins below carry exactly the members the two
// policy classes touch, with the same names and meaning:
//   openMVG::matching::IndMatch{i_, j_}, IndMatches, PairWiseMatches; openMVG::Pair, Pair_Set;
//   openMVG::features::AKAZE_Binary_Regions with Features() / Descriptors() / DescriptorRawData() /
//   RegionCount() / GetRegionPosition(); coloc::FeatureMap, DetectorOptions, MatcherOptions
//   (include/coloc/colocData.hpp:20-42) and the slice of coloc::colocData that matchSceneWithMap / setupTracks read
//   (regions, mapRegions, scene landmarks, mapRegionIdx); openMVG::Vec2 / Vec3.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <utility>
#include <vector>

#ifndef COLOC_HIP_WITH_OPENMVG

namespace openMVG {
using IndexT = uint32_t;
using Pair = std::pair<IndexT, IndexT>;
using Pair_Set = std::set<Pair>;

// openMVG exhaustivePairs(N): all (i, j) with i < j -- what coloc::Utils::handlePairs returns
// (include/coloc/colocUtils.hpp:58-61)
inline Pair_Set exhaustivePairs(const size_t n)
{
    Pair_Set pairs;
    for (IndexT i = 0; i < static_cast<IndexT>(n); ++i)
        for (IndexT j = i + 1; j < static_cast<IndexT>(n); ++j) pairs.insert({ i, j });
    return pairs;
}

namespace matching {
struct IndMatch {
    IndexT i_ = 0, j_ = 0;
    IndMatch() = default;
    IndMatch(IndexT i, IndexT j) : i_(i), j_(j) {}
    bool operator==(const IndMatch& o) const { return i_ == o.i_ && j_ == o.j_; }
};
using IndMatches = std::vector<IndMatch>;
using PairWiseMatches = std::map<Pair, IndMatches>;
} // namespace matching

namespace features {
struct SIOPointFeature {
    float x_ = 0, y_ = 0, scale_ = 0, orientation_ = 0;
    SIOPointFeature() = default;
    SIOPointFeature(float x, float y, float s, float o) : x_(x), y_(y), scale_(s), orientation_(o) {}
    float x() const { return x_; }
    float y() const { return y_; }
    float scale() const { return scale_; }
    float orientation() const { return orientation_; }
};
template <typename T, size_t N>
using Descriptor = std::array<T, N>;
// what Regions::GetRegionsPositions() hands out (RobustMatcher.hpp:297-298): positions only
struct PointFeature {
    float x_ = 0, y_ = 0;
    PointFeature() = default;
    PointFeature(float x, float y) : x_(x), y_(y) {}
    float x() const { return x_; }
    float y() const { return y_; }
};
using PointFeatures = std::vector<PointFeature>;

// Binary_Regions<SIOPointFeature, 64>
class AKAZE_Binary_Regions {
public:
    using FeatureT = SIOPointFeature;
    using DescriptorT = Descriptor<unsigned char, 64>;
    std::vector<FeatureT>& Features() { return feats_; }
    const std::vector<FeatureT>& Features() const { return feats_; }
    std::vector<DescriptorT>& Descriptors() { return descs_; }
    const std::vector<DescriptorT>& Descriptors() const { return descs_; }
    const void* DescriptorRawData() const { return descs_.data(); }
    size_t RegionCount() const { return feats_.size(); }
    std::array<double, 2> GetRegionPosition(size_t i) const { return { feats_[i].x(), feats_[i].y() }; }
    PointFeatures GetRegionsPositions() const
    {
        PointFeatures p;
        p.reserve(feats_.size());
        for (const FeatureT& f : feats_) p.emplace_back(f.x(), f.y());
        return p;
    }
private:
    std::vector<FeatureT> feats_;
    std::vector<DescriptorT> descs_;
};
} // namespace features
} // namespace openMVG

namespace openMVG {
// fixed-size vectors and the landmark container: what colocData::scene exposes to Localizer::setupTracks
// (sfm::SfM_Data::GetLandmarks().at(id).X, include/coloc/Localizer.hpp:66)
template <size_t N>
struct VecN {
    std::array<double, N> v{};
    VecN() = default;
    VecN(double a, double b) { static_assert(N == 2, "Vec2"); v = { a, b }; }
    VecN(double a, double b, double c) { static_assert(N == 3, "Vec3"); v = { a, b, c }; }
    double& operator()(size_t i) { return v[i]; }
    double operator()(size_t i) const { return v[i]; }
    double& operator[](size_t i) { return v[i]; }
    double operator[](size_t i) const { return v[i]; }
};
using Vec2 = VecN<2>;
using Vec3 = VecN<3>;
namespace sfm {
struct Landmark { Vec3 X; };
using Landmarks = std::map<IndexT, Landmark>;
struct SfM_Data {
    Landmarks structure;
    const Landmarks& GetLandmarks() const { return structure; }
};
} // namespace sfm
} // namespace openMVG

namespace coloc {
using FeatureMap = std::map<openMVG::IndexT, std::unique_ptr<openMVG::features::AKAZE_Binary_Regions>>;
using Scene = openMVG::sfm::SfM_Data;

struct DetectorOptions {   // include/coloc/colocData.hpp:29-36
    float scale_factor;
    uint8_t scale_levels;
    unsigned int width;
    unsigned int height;
    unsigned int maxkp;
    uint8_t thresh;
};
struct MatcherOptions {    // include/coloc/colocData.hpp:38-42
    float distRatio;
    int thresh;
    unsigned int maxkp;
};
// the members GPUMatcher::matchSceneWithMap and Localizer::setupTracks read (include/coloc/colocData.hpp:47-56)
struct colocData {
    FeatureMap regions;
    Scene scene;
    std::unique_ptr<openMVG::features::AKAZE_Binary_Regions> mapRegions;
    std::vector<openMVG::IndexT> mapRegionIdx;
};
namespace Utils {
inline openMVG::Pair_Set handlePairs(int numImages) { return openMVG::exhaustivePairs(numImages); }
} // namespace Utils
} // namespace coloc

#endif // COLOC_HIP_WITH_OPENMVG

#ifndef COLOC_HIP_HAVE_KEYPOINT_H
// include/coloc/Keypoint.h:155-163 (20 bytes; layout shared with clc_keypoint)
struct Keypoint {
    int32_t x;
    int32_t y;
    uint8_t score;
    float angle;
    uint8_t scale;
    Keypoint() {}
    Keypoint(const int32_t _x, const int32_t _y, const uint8_t _score) : x(_x), y(_y), score(_score) {}
};
#endif

namespace coloc {
// cv::KeyPoint stand-in for GPUDetector::converted_kps (GPUDetector.hpp:35) when OpenCV is absent
struct HipKeyPoint {
    float x = 0, y = 0, size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
// cv::DMatch(queryIdx, trainIdx, distance) as filled at GPUMatcher.hpp:218 when OpenCV is absent
struct HipDMatch {
    int queryIdx, trainIdx;
    float distance;
    HipDMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), distance(d) {}
};
} // namespace coloc
