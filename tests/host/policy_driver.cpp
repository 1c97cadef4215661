// policy_driver.cpp -- exercises HIPDetector<bool> / HIPMatcher<bool> exactly the way ColoC /
// DiskInterface use GPUDetector / GPUMatcher (reference include/coloc/coloc.hpp:162,197,219,287,323;
// InterfaceDisk.hpp:15) and dumps the results as raw files for tests/test_gpu_policy.py to compare
// with the oracle.  usage: policy_driver <dir> <ncams> <width> <height> <maxkp>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#include <fstream>
#include <string>

#include "HIPDetector.hpp"
#include "HIPMatcher.hpp"

using namespace openMVG;
using namespace openMVG::matching;

template <typename T, template <class> class ProcessorType>
class FeatureDetector : public ProcessorType<T> {   // policy host, reference FeatureDetector.hpp:21-32
public:
    explicit FeatureDetector(coloc::DetectorOptions& opts) : ProcessorType<T>(opts) {}
    T detectFeaturesFile(unsigned int idx, coloc::FeatureMap& regions, std::string& imageName)
    {
        return ProcessorType<T>::detectFeaturesFile(idx, regions, imageName);
    }
};
template <typename T, template <class> class ProcessorType>
class FeatureMatcher : public ProcessorType<T> {    // reference FeatureMatcher.hpp:23-34
public:
    explicit FeatureMatcher(coloc::MatcherOptions& opts) : ProcessorType<T>(opts) {}
    bool computeMatches(coloc::FeatureMap& regions, PairWiseMatches& putativeMatches)
    {
        return ProcessorType<T>::computeMatches(regions, putativeMatches);
    }
};

static void dump(const std::string& path, const void* p, size_t bytes)
{
    std::ofstream f(path, std::ios::binary);
    f.write(static_cast<const char*>(p), static_cast<std::streamsize>(bytes));
}
static void dump_matches(const std::string& path, const IndMatches& m)
{
    std::vector<uint32_t> flat;
    for (const auto& e : m) { flat.push_back(e.i_); flat.push_back(e.j_); }
    dump(path, flat.data(), flat.size() * 4);
}

int main(int argc, char** argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: %s dir ncams width height maxkp\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    const int ncams = std::atoi(argv[2]);
    coloc::DetectorOptions dopts{ 1.2f, 8, static_cast<unsigned>(std::atoi(argv[3])), static_cast<unsigned>(std::atoi(argv[4])),
                                  static_cast<unsigned>(std::atoi(argv[5])), 40 };           // coloc_node.cpp:76-81
    coloc::MatcherOptions mopts{ 0.8f, 60, static_cast<unsigned>(std::atoi(argv[5])) };      // coloc_node.cpp:83-85
    FeatureDetector<bool, coloc::HIPDetector> detector(dopts);
    FeatureMatcher<bool, coloc::HIPMatcher> matcher(mopts);

    coloc::colocData data;
    for (int c = 0; c < ncams; ++c) {
        std::string name = dir + "/img" + std::to_string(c) + ".pgm";
        if (detector.detectFeaturesFile(c, data.regions, name) != EXIT_SUCCESS) { std::fprintf(stderr, "detect failed\n"); return 1; }
        dump(dir + "/kps" + std::to_string(c) + ".bin", detector.kps.data(), detector.kps.size() * sizeof(Keypoint));
        dump(dir + "/desc" + std::to_string(c) + ".bin", data.regions[c]->DescriptorRawData(), data.regions[c]->RegionCount() * 64);
        dump(dir + "/feat" + std::to_string(c) + ".bin", data.regions[c]->Features().data(), data.regions[c]->RegionCount() * 16);
    }
    // the topic entry (GPUDetector.hpp:188-212) with a cv_bridge-shaped holder: same regions as the file entry
    {
        struct Img { unsigned char* data; int cols, rows; };
        struct Holder { Img image; };
        std::vector<uint8_t> pix;
        int w = 0, h = 0;
        std::string name = dir + "/img0.pgm";
        if (!coloc::hip_detail::read_pgm(name, pix, w, h)) return 1;
        Holder holder{ { pix.data(), w, h } };
        coloc::FeatureMap topic;
        detector.detectFeaturesTopic(0, topic, &holder);
        if (topic.count(0) == 0 || topic[0]->RegionCount() != data.regions[0]->RegionCount() || !detector.converted_kps.empty() ||
            std::memcmp(topic[0]->DescriptorRawData(), data.regions[0]->DescriptorRawData(), topic[0]->RegionCount() * 64) != 0 ||
            std::memcmp(topic[0]->Features().data(), data.regions[0]->Features().data(), topic[0]->RegionCount() * 16) != 0) {
            std::fprintf(stderr, "detectFeaturesTopic differs from detectFeaturesFile\n");
            return 1;
        }
    }
    // bad file -> EXIT_FAILURE (true), nothing inserted
    {
        std::string bad = dir + "/does_not_exist.pgm";
        coloc::FeatureMap tmp;
        if (detector.detectFeaturesFile(0, tmp, bad) != EXIT_FAILURE || !tmp.empty()) { std::fprintf(stderr, "bad-file convention broken\n"); return 1; }
    }
    // initMap: all pairs (coloc.hpp:162)
    PairWiseMatches putative;
    if (matcher.computeMatches(data.regions, putative) != EXIT_SUCCESS) return 1;
    for (const auto& kv : putative)
        dump_matches(dir + "/pair_" + std::to_string(kv.first.first) + "_" + std::to_string(kv.first.second) + ".bin", kv.second);
    // interPoseEstimator: one pair (coloc.hpp:287)
    IndMatches one;
    matcher.computeMatchesPair({ 0, 1 }, data.regions, one);
    dump_matches(dir + "/single_0_1.bin", one);
    // map tracking (coloc.hpp:196-198, 219): map = camera 0's regions, query = camera 1
    data.mapRegions.reset(new features::AKAZE_Binary_Regions(*data.regions[0]));
    matcher.setMapData(static_cast<int>(data.mapRegions->RegionCount()), const_cast<void*>(data.mapRegions->DescriptorRawData()));
    int drone = 1;
    IndMatches mapMatches;
    matcher.matchSceneWithMap(drone, data, mapMatches);
    dump_matches(dir + "/map_1.bin", mapMatches);
    // map <-> map (coloc.hpp:323), thr 60
    IndMatches common;
    matcher.matchMapFeatures(data.regions[0], data.regions[1], common);
    dump_matches(dir + "/mapmap_0_1.bin", common);
    // ---- regions edited IN PLACE after the detector stored (and published) them, in the policy classes' DEFAULT mode: every match
    // entry must answer for the rows the block holds NOW.  One byte of a middle row that no sampled row covers, then
    // a stretch of rows copied over from camera 1 -- each followed by all the match entry points; the handle the detector got says when
    // its publication has died.
    {
        coloc::FeatureMap fresh;
        std::string name = dir + "/img0.pgm";
        if (detector.detectFeaturesFile(0, fresh, name) != EXIT_SUCCESS) return 1;
        const clc_desc_handle published = detector.lastPublished();
        if (!published.host || published.host != fresh[0]->DescriptorRawData() || !clc_desc_handle_live(&published)) {
            std::fprintf(stderr, "the detector did not publish regions[0] (host %p)\n", published.host);
            return 1;
        }
        // regions 1 stays as the detector stored it (still published); regions 0 is the fresh block
        data.regions[0] = std::move(fresh[0]);
        const size_t n0 = data.regions[0]->RegionCount(), n1 = data.regions[1]->RegionCount();
        IndMatches a;
        matcher.computeMatchesPair({ 0, 1 }, data.regions, a);                       // unchanged block: served from the device
        dump_matches(dir + "/edit0_0_1.bin", a);
        if (!clc_desc_handle_live(&published)) { std::fprintf(stderr, "an unchanged block lost its publication\n"); return 1; }
        const size_t mid = n0 / 2 + 3;
        data.regions[0]->Descriptors()[mid][20] ^= 0x04;
        if (n1 > 8) data.regions[0]->Descriptors()[mid] = data.regions[1]->Descriptors()[7];   // a row that now matches camera 1's row 7 exactly
        matcher.computeMatchesPair({ 0, 1 }, data.regions, a);
        dump_matches(dir + "/edit1_0_1.bin", a);
        if (clc_desc_handle_live(&published)) { std::fprintf(stderr, "an edited block kept its publication\n"); return 1; }
        dump(dir + "/edit1_desc0.bin", data.regions[0]->DescriptorRawData(), n0 * 64);
        // publish it again through a second detect of the same frame, then overwrite a stretch and go through the other entries
        coloc::FeatureMap again;
        if (detector.detectFeaturesFile(0, again, name) != EXIT_SUCCESS) return 1;
        data.regions[0] = std::move(again[0]);
        const size_t k = std::min<size_t>(n1, std::min<size_t>(n0, 300)) / 2;
        for (size_t i = 0; i < k; ++i) data.regions[0]->Descriptors()[n0 / 3 + i] = data.regions[1]->Descriptors()[i];
        dump(dir + "/edit2_desc0.bin", data.regions[0]->DescriptorRawData(), n0 * 64);
        PairWiseMatches all;
        if (matcher.computeMatches(data.regions, all) != EXIT_SUCCESS) return 1;
        dump_matches(dir + "/edit2_0_1.bin", all.count({ 0, 1 }) ? all[{ 0, 1 }] : IndMatches());
        int d0 = 0;
        IndMatches mm;
        matcher.matchSceneWithMap(d0, data, mm);                                     // query = the edited regions 0, map = the old camera 0
        dump_matches(dir + "/edit2_map_0.bin", mm);
        IndMatches cm;
        matcher.matchMapFeatures(data.regions[1], data.regions[0], cm);              // the edited block as the TRAIN side
        dump_matches(dir + "/edit2_mapmap_1_0.bin", cm);
    }
    std::printf("ok %d cams, %zu pairs with matches\n", ncams, putative.size());
    return 0;
}
