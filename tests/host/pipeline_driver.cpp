// pipeline_driver.cpp -- the four HIP policy classes chained the way ColoC chains the reference's (include/coloc/coloc.hpp):
//   processImages      :150-163   detector.detectFeaturesFile(i, data.regions, file)            per camera
//   initMap            :162-169   matcher.computeMatches(regions, putative); robustMatcher.filterMatches(regions, putative, geometric, poses)
//   intraPoseEstimator :197-223   matcher.setMapData(...); matcher.matchSceneWithMap(id, data, mapMatches);
//                                 localizer.localizeImage(id, pose, data, cov, rmse, mapMatches, inliers)
//   interPoseEstimator :323-326   matcher.matchMapFeatures(map, interMap, common); robustMatcher.matchMaps(map, interMap, common, poseDiff, rotDiff)
// on frames rendered by tests/test_gpu_pipeline_host.py.  The map (3-D points under camera 0's features) comes from the test,
// which knows the scene: the driver runs twice -- "features" dumps camera 0's feature positions, "run" does everything.
// usage: pipeline_driver features|run <dir> <width> <height> <focal> <ppx> <ppy>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "HIPDetector.hpp"
#include "HIPMatcher.hpp"
#include "HIPLocalizer.hpp"
#include "HIPRobustMatcher.hpp"

using namespace openMVG;
using namespace openMVG::matching;

static void dump(const std::string& path, const std::vector<double>& v)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), static_cast<std::streamsize>(v.size() * 8));
}
static std::vector<double> slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<double> v(f ? static_cast<size_t>(f.tellg()) / 8 : 0);
    if (f) { f.seekg(0); f.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(v.size() * 8)); }
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 8) { std::fprintf(stderr, "usage: %s features|run dir w h focal ppx ppy\n", argv[0]); return 2; }
    const std::string mode = argv[1], dir = argv[2];
    const unsigned w = std::atoi(argv[3]), h = std::atoi(argv[4]);
    const double f = std::atof(argv[5]), ppx = std::atof(argv[6]), ppy = std::atof(argv[7]);
    coloc::DetectorOptions dopts{ 1.2f, 8, w, h, 12000, 40 };              // coloc_node.cpp:76-81
    coloc::MatcherOptions mopts{ 0.8f, 60, 12000 };                        // coloc_node.cpp:83-85
    Mat3 K; K(0, 0) = f; K(1, 1) = f; K(0, 2) = ppx; K(1, 2) = ppy; K(2, 2) = 1.0;
    coloc::colocParams params({ K, K }, { Vec3(0, 0, 0), Vec3(0, 0, 0) }, 'E', { (size_t)w, (size_t)h }, dir, dopts, mopts);

    coloc::HIPDetector<bool> detector(dopts);
    coloc::colocData data;
    const int ncams = mode == "features" ? 1 : 2;
    for (int c = 0; c < ncams; ++c) {
        std::string name = dir + "/cam" + std::to_string(c) + ".pgm";
        if (detector.detectFeaturesFile(c, data.regions, name) != EXIT_SUCCESS) { std::fprintf(stderr, "detect failed\n"); return 1; }
    }
    if (mode == "features") {
        std::vector<double> out;
        for (size_t i = 0; i < data.regions[0]->RegionCount(); ++i) {
            const auto p = data.regions[0]->GetRegionPosition(i);
            out.push_back(p[0]); out.push_back(p[1]);
        }
        dump(dir + "/feat0.bin", out);
        return 0;
    }
    std::vector<double> out;
    // ---- initMap: putative matches of the pair, geometric filter, relative pose
    coloc::HIPMatcher<bool> matcher(mopts);
    PairWiseMatches putative, geometric;
    if (matcher.computeMatches(data.regions, putative) != EXIT_SUCCESS) { std::fprintf(stderr, "computeMatches failed\n"); return 1; }
    coloc::HIPRobustMatcher robust(params);
    coloc::InterPoseMap relativePoses;
    robust.filterMatches(data.regions, putative, geometric, relativePoses);
    const Pair pr(0, 1);
    out.push_back((double)data.regions[0]->RegionCount());
    out.push_back((double)data.regions[1]->RegionCount());
    out.push_back((double)putative[pr].size());
    out.push_back((double)geometric[pr].size());
    const geometry::Pose3& rel = relativePoses[pr].relativePose;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(rel.rotation()(i, j));
    for (int i = 0; i < 3; ++i) out.push_back(rel.center()[i]);
    // ---- intraPoseEstimator: camera 0's features are the map (3-D points from the test), camera 1 is localized against it
    const std::vector<double> X = slurp(dir + "/map_xyz.bin");            // 3 x RegionCount(0)
    if (X.size() != 3 * data.regions[0]->RegionCount()) { std::fprintf(stderr, "map size mismatch\n"); return 1; }
    data.mapRegions.reset(new features::AKAZE_Binary_Regions);
    for (size_t i = 0; i < data.regions[0]->RegionCount(); ++i) {
        data.mapRegions->Features().push_back(data.regions[0]->Features()[i]);
        data.mapRegions->Descriptors().push_back(data.regions[0]->Descriptors()[i]);
        data.scene.structure[(IndexT)i].X = Vec3(X[3 * i], X[3 * i + 1], X[3 * i + 2]);
        data.mapRegionIdx.push_back((IndexT)i);
    }
    matcher.setMapData((int)data.mapRegions->RegionCount(), const_cast<void*>(static_cast<const void*>(data.mapRegions->DescriptorRawData())));
    int droneId = 1;
    IndMatches mapMatches;
    matcher.matchSceneWithMap(droneId, data, mapMatches);
    coloc::HIPLocalizer localizer(params);
    geometry::Pose3 pose;
    coloc::Cov6 cov;
    float rmse = -1.0f;
    std::vector<uint32_t> inliers;
    const bool status = localizer.localizeImage(droneId, pose, data, cov, rmse, mapMatches, inliers);
    out.push_back((double)mapMatches.size());
    out.push_back(status ? 1.0 : 0.0);
    out.push_back((double)inliers.size());
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(pose.rotation()(i, j));
    for (int i = 0; i < 3; ++i) out.push_back(pose.center()[i]);
    out.push_back(rmse);
    // ---- interPoseEstimator, the map-to-map part (coloc.hpp:323-326): the global map against a second map -- here camera 1's
    // features -- through matchMapFeatures (thr 60), then matchMaps with the displacement between the two views
    std::unique_ptr<features::AKAZE_Binary_Regions> interMapRegions(new features::AKAZE_Binary_Regions);
    for (size_t i = 0; i < data.regions[1]->RegionCount(); ++i) {
        interMapRegions->Features().push_back(data.regions[1]->Features()[i]);
        interMapRegions->Descriptors().push_back(data.regions[1]->Descriptors()[i]);
    }
    std::vector<IndMatch> commonFeatures;
    matcher.matchMapFeatures(data.mapRegions, interMapRegions, commonFeatures);
    const std::vector<IndMatch> before = commonFeatures;
    Vec3 poseDiff = pose.center();
    Mat3 rotDiff = pose.rotation();
    const bool mm = robust.matchMaps(data.mapRegions, interMapRegions, commonFeatures, poseDiff, rotDiff);
    bool kept = before.size() == commonFeatures.size();
    for (size_t i = 0; kept && i < before.size(); ++i) kept = before[i] == commonFeatures[i];
    out.push_back((double)commonFeatures.size());
    out.push_back(mm ? 1.0 : 0.0);
    out.push_back(kept ? 1.0 : 0.0);
    dump(dir + "/pipeline_out.bin", out);
    return 0;
}
