// bench_policy.cpp -- times the drop-in path itself: HIPDetector / HIPMatcher / HIPLocalizer driven from C++ in the order of
// ColoC::mainThread (reference include/coloc/coloc.hpp:111-148), the same three spans the reference prints around its own calls:
//   "Detection in milliseconds"  :129-136  colocInterface.processImageSingle(i) -> detector.detect*(i, data.regions, image)
//   "Matching in milliseconds"   :161-164  matcher.computeMatches(data.regions, data.putativeMatches)        (initMap, once per map)
//   "Tracking in milliseconds"   :217-221  matcher.matchSceneWithMap(droneId, data, mapMatches)              (per frame)
//   "PNP in ms"                  :222-225  localizer.localizeImage(droneId, pose, data, cov, rmse, mapMatches, inliers)
// What this measures is what coloc_node would link against: host buffers in, OpenMVG-shaped regions / matches / pose out, every copy,
// allocation and synchronisation of the policy classes included -- next to bench.py's device-resident headline, never instead of it.
// Images come from memory (what detectFeaturesTopic hands over, GPUDetector.hpp:188-212): the reference's file variant also times
// cv::imread, which is not ours to speed up.  Frames: camera 1 and camera 0 alternate (mainThread's `for i < 2`), the map is camera 0's
// first frame with the 3-D points the harness (tests/test_gpu_policy_bench.py, bench.py) computed for its features.
// usage: bench_policy <dir> <width> <height> <focal> <ppx> <ppy> <frames> <warmup> [maxkp]
// prints ONE line: POLICY {json}
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "HIPDetector.hpp"
#include "HIPMatcher.hpp"
#include "HIPLocalizer.hpp"

using namespace openMVG;
using namespace openMVG::matching;
using clk = std::chrono::steady_clock;

static std::vector<double> slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<double> v(f ? static_cast<size_t>(f.tellg()) / 8 : 0);
    if (f) { f.seekg(0); f.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(v.size() * 8)); }
    return v;
}
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static double pct(std::vector<double> v, double p)
{
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, static_cast<size_t>(p * (v.size() - 1) + 0.5))];
}
static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 9) { std::fprintf(stderr, "usage: %s dir w h focal ppx ppy frames warmup [maxkp]\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    const unsigned w = std::atoi(argv[2]), h = std::atoi(argv[3]);
    const double f = std::atof(argv[4]), ppx = std::atof(argv[5]), ppy = std::atof(argv[6]);
    const int frames = std::atoi(argv[7]), warmup = std::atoi(argv[8]);
    const unsigned maxkp = argc > 9 ? std::atoi(argv[9]) : 12000;
    coloc::DetectorOptions dopts{ 1.2f, 8, w, h, maxkp, 40 };              // coloc_node.cpp:76-81
    coloc::MatcherOptions mopts{ 0.8f, 60, maxkp };                         // coloc_node.cpp:83-85
    Mat3 K; K(0, 0) = f; K(1, 1) = f; K(0, 2) = ppx; K(1, 2) = ppy; K(2, 2) = 1.0;
    coloc::colocParams params({ K, K }, { Vec3(0, 0, 0), Vec3(0, 0, 0) }, 'E', { (size_t)w, (size_t)h }, dir, dopts, mopts);

    std::vector<uint8_t> img[2];
    for (int c = 0; c < 2; ++c) {
        int iw = 0, ih = 0;
        if (!coloc::hip_detail::read_pgm(dir + "/cam" + std::to_string(c) + ".pgm", img[c], iw, ih) || (unsigned)iw != w || (unsigned)ih != h) {
            std::fprintf(stderr, "cannot read cam%d.pgm\n", c);
            return 1;
        }
    }
    coloc::HIPDetector<bool> detector(dopts);
    coloc::HIPMatcher<bool> matcher(mopts);
    coloc::HIPLocalizer localizer(params);
    if (const char* e = std::getenv("BENCH_POLICY_PUBLISH")) {               // A/B: 0 = every match call uploads (the reference's behaviour)
        if (e[0] == '0') { detector.publishRegions(false); matcher.usePublishedRegions(false); }
        if (e[0] == 't') matcher.trustPublishedRegions(true);
    }
    coloc::colocData data;
    // the map: camera 0's first frame + the 3-D points under its features (initMap's outcome, coloc.hpp:150-194)
    if (detector.detectFeaturesImage(0, data.regions, img[0].data(), (int)w, (int)h) != EXIT_SUCCESS) { std::fprintf(stderr, "detect failed\n"); return 1; }
    const std::vector<double> X = slurp(dir + "/map_xyz.bin");
    if (X.size() != 3 * data.regions[0]->RegionCount()) {
        std::fprintf(stderr, "map_xyz.bin holds %zu points, camera 0 has %zu features\n", X.size() / 3, data.regions[0]->RegionCount());
        return 1;
    }
    data.mapRegions.reset(new features::AKAZE_Binary_Regions);
    for (size_t i = 0; i < data.regions[0]->RegionCount(); ++i) {
        data.mapRegions->Features().push_back(data.regions[0]->Features()[i]);
        data.mapRegions->Descriptors().push_back(data.regions[0]->Descriptors()[i]);
        data.scene.structure[(IndexT)i].X = Vec3(X[3 * i], X[3 * i + 1], X[3 * i + 2]);
        data.mapRegionIdx.push_back((IndexT)i);
    }
    matcher.setMapData((int)data.mapRegions->RegionCount(), const_cast<void*>(static_cast<const void*>(data.mapRegions->DescriptorRawData())));

    std::vector<double> t_detect, t_match, t_pose, t_frame, t_pair;
    size_t kp[2] = { 0, 0 }, n_map[2] = { 0, 0 }, n_inl[2] = { 0, 0 };
    uint64_t digest[2] = { 0, 0 };
    int failures = 0;
    for (int it = 0; it < warmup + frames; ++it) {
        int droneId = (it & 1) ? 0 : 1;
        const clk::time_point a = clk::now();
        const bool det = detector.detectFeaturesImage(static_cast<uint8_t>(droneId), data.regions, img[droneId].data(), (int)w, (int)h);
        const clk::time_point b = clk::now();
        IndMatches mapMatches;
        matcher.matchSceneWithMap(droneId, data, mapMatches);
        const clk::time_point c = clk::now();
        geometry::Pose3 pose;
        coloc::Cov6 cov;
        float rmse = -1.0f;
        std::vector<uint32_t> inliers;
        localizer.seed = 1;                                                 // the same sampler stream every frame: the same pose every frame
        const bool loc = localizer.localizeImage(droneId, pose, data, cov, rmse, mapMatches, inliers);
        const clk::time_point d = clk::now();
        if (det != EXIT_SUCCESS || loc != EXIT_SUCCESS) ++failures;
        // what the frame produced, folded: the same digest every frame of a camera (and equal to the device-pointer path's, checked by the harness)
        uint64_t dg = fnv(data.regions[droneId]->DescriptorRawData(), data.regions[droneId]->RegionCount() * 64);
        dg = fnv(data.regions[droneId]->Features().data(), data.regions[droneId]->RegionCount() * 16, dg);
        dg = fnv(mapMatches.data(), mapMatches.size() * sizeof(IndMatch), dg);
        dg = fnv(inliers.data(), inliers.size() * 4, dg);
        for (int i = 0; i < 3; ++i) { const double v = pose.center()[i]; dg = fnv(&v, 8, dg); }
        if (it >= warmup) {
            if (digest[droneId] == 0) digest[droneId] = dg;
            else if (digest[droneId] != dg) ++failures;
            t_detect.push_back(us(a, b)); t_match.push_back(us(b, c)); t_pose.push_back(us(c, d)); t_frame.push_back(us(a, d));
            kp[droneId] = data.regions[droneId]->RegionCount(); n_map[droneId] = mapMatches.size(); n_inl[droneId] = inliers.size();
        }
    }
    // the pair match of initMap (coloc.hpp:161-164) on the two cameras' last frames
    size_t n_pair = 0;
    for (int it = 0; it < warmup / 4 + frames / 4 + 4; ++it) {
        PairWiseMatches putative;
        const clk::time_point a = clk::now();
        if (matcher.computeMatches(data.regions, putative) != EXIT_SUCCESS) ++failures;
        const clk::time_point b = clk::now();
        if (it >= warmup / 4) t_pair.push_back(us(a, b));
        n_pair = putative.count({ 0, 1 }) ? putative[{ 0, 1 }].size() : 0;
    }
    // the descriptor files of the last frames, for the harness to compare with the device-pointer path
    for (int c = 0; c < 2; ++c) {
        std::ofstream o(dir + "/policy_desc" + std::to_string(c) + ".bin", std::ios::binary);
        o.write(static_cast<const char*>(data.regions[c]->DescriptorRawData()), static_cast<std::streamsize>(data.regions[c]->RegionCount() * 64));
        std::ofstream k(dir + "/policy_kps" + std::to_string(c) + ".bin", std::ios::binary);
        k.write(reinterpret_cast<const char*>(data.regions[c]->Features().data()), static_cast<std::streamsize>(data.regions[c]->RegionCount() * 16));
    }
    std::printf("POLICY {\"frames\": %d, \"warmup\": %d, \"width\": %u, \"height\": %u, \"keypoints\": [%zu, %zu], \"map_points\": %zu, "
                "\"map_matches\": [%zu, %zu], \"pose_inliers\": [%zu, %zu], \"pair_matches\": %zu, "
                "\"detect_us\": %.1f, \"detect_us_p95\": %.1f, \"match_us\": %.1f, \"match_us_p95\": %.1f, \"pose_us\": %.1f, \"pose_us_p95\": %.1f, "
                "\"frame_us\": %.1f, \"frame_us_p95\": %.1f, \"pair_match_us\": %.1f, \"pair_match_us_p95\": %.1f, \"failures\": %d, "
                "\"same_results_every_frame\": %s}\n",
                frames, warmup, w, h, kp[0], kp[1], data.mapRegions->RegionCount(), n_map[0], n_map[1], n_inl[0], n_inl[1], n_pair,
                pct(t_detect, 0.5), pct(t_detect, 0.95), pct(t_match, 0.5), pct(t_match, 0.95), pct(t_pose, 0.5), pct(t_pose, 0.95),
                pct(t_frame, 0.5), pct(t_frame, 0.95), pct(t_pair, 0.5), pct(t_pair, 0.95), failures, failures == 0 ? "true" : "false");
    return failures == 0 ? 0 : 3;
}
