// bench_policy.cpp -- times the drop-in path itself: HIPDetector / HIPMatcher / HIPLocalizer driven from C++ in the order of
// ColoC::mainThread (reference include/coloc/coloc.hpp:111-148), the same three spans the reference prints around its own calls:
//   "Detection in milliseconds"  :129-136  colocInterface.processImageSingle(i) -> detector.detect*(i, data.regions, image)
//   "Matching in milliseconds"   :161-164  matcher.computeMatches(data.regions, data.putativeMatches)        (initMap, once per map)
//   "Tracking in milliseconds"   :217-221  matcher.matchSceneWithMap(droneId, data, mapMatches)              (per frame)
//   "PNP in ms"                  :222-225  localizer.localizeImage(droneId, pose, data, cov, rmse, mapMatches, inliers)
// What this measures is what coloc_node would link against: host buffers in, OpenMVG-shaped regions / matches / pose out, every copy,
// allocation and synchronisation of the policy classes included -- next to bench.py's device-resident headline, never instead of it.
// Images come from memory (what detectFeaturesTopic hands over, GPUDetector.hpp:188-212): the reference's file variant also times
// cv::imread, which is not ours to speed up.  Frames: camera 1 and camera 0 alternate (mainThread's `for i < 2`); the map is a THIRD view
// of the scene (cam_map.pgm: a keyframe's features) with the 3-D points the harness (tests/test_gpu_policy_bench.py, bench.py) computed
// for its features, so that both cameras track the map through a few hundred matches like a real frame does.
// usage: bench_policy <dir> <width> <height> <focal> <ppx> <ppy> <frames> <warmup> [maxkp]
// prints ONE line: POLICY {json}
#include <algorithm>
#include <chrono>
#include <cmath>
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

    std::vector<uint8_t> img[3];
    const char* names[3] = { "/cam0.pgm", "/cam1.pgm", "/cam_map.pgm" };
    for (int c = 0; c < 3; ++c) {
        int iw = 0, ih = 0;
        if (!coloc::hip_detail::read_pgm(dir + names[c], img[c], iw, ih) || (unsigned)iw != w || (unsigned)ih != h) {
            std::fprintf(stderr, "cannot read %s\n", names[c]);
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
    // the map: the keyframe's features + the 3-D points under them (initMap's outcome, coloc.hpp:150-194)
    {
        coloc::FeatureMap key;
        if (detector.detectFeaturesImage(0, key, img[2].data(), (int)w, (int)h) != EXIT_SUCCESS) { std::fprintf(stderr, "detect failed\n"); return 1; }
        const std::vector<double> X = slurp(dir + "/map_xyz.bin");
        if (X.size() != 3 * key[0]->RegionCount()) {
            std::fprintf(stderr, "map_xyz.bin holds %zu points, the keyframe has %zu features\n", X.size() / 3, key[0]->RegionCount());
            return 1;
        }
        data.mapRegions.reset(new features::AKAZE_Binary_Regions);
        for (size_t i = 0; i < key[0]->RegionCount(); ++i) {
            data.mapRegions->Features().push_back(key[0]->Features()[i]);
            data.mapRegions->Descriptors().push_back(key[0]->Descriptors()[i]);
            data.scene.structure[(IndexT)i].X = Vec3(X[3 * i], X[3 * i + 1], X[3 * i + 2]);
            data.mapRegionIdx.push_back((IndexT)i);
        }
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
    // where detect_us goes: HIPDetector::detectFeaturesImage's own steps spelled out on a context of its own, each between two clock reads
    // (the C entry: image into the pinned block, enqueue, the frame's synchronisation | keypoints + features | freeing the last frame's
    //  regions, room for the new ones | the descriptors' one copy, folded and published)
    std::vector<double> t_view;
    double bd[4] = { 0, 0, 0, 0 };
    {
        clc_detector_opts d{ dopts.scale_factor, dopts.scale_levels, dopts.width, dopts.height, dopts.maxkp, dopts.thresh };
        clc_ctx* raw = nullptr;
        if (clc_ctx_create(0, &d, nullptr, &raw) == CLC_OK) {
            coloc::FeatureMap fm;
            std::vector<Keypoint> kv;
            float ls[256];
            for (int l = 0; l < 256; ++l) ls[l] = static_cast<float>(std::pow(static_cast<double>(1.2f), static_cast<double>(l)));
            for (int it = 0; it < warmup + frames; ++it) {
                const clc_keypoint* pk = nullptr; const uint8_t* pd = nullptr; int n = 0, found = 0;
                const clk::time_point a = clk::now();
                if (clc_detect_and_describe_view(raw, img[(it & 1) ? 0 : 1].data(), w, h, &pk, &pd, &n, &found) != CLC_OK) ++failures;
                const clk::time_point b = clk::now();
                kv.resize((size_t)n);
                if (n > 0) std::memcpy(static_cast<void*>(kv.data()), pk, (size_t)n * sizeof(Keypoint));
                const clk::time_point c = clk::now();
                fm[0] = std::unique_ptr<features::AKAZE_Binary_Regions>(new features::AKAZE_Binary_Regions);
                fm[0]->Features().resize((size_t)n);
                fm[0]->Descriptors().resize((size_t)n);
                for (int i = 0; i < n; ++i) {
                    const float sc = ls[kv[(size_t)i].scale];
                    fm[0]->Features()[(size_t)i] = { sc * (float)kv[(size_t)i].x, sc * (float)kv[(size_t)i].y, 7.0f * sc, kv[(size_t)i].angle };
                }
                const clk::time_point e = clk::now();
                if (n > 0 && clc_detect_store_descriptors(raw, fm[0]->Descriptors().data(), n, nullptr) != CLC_OK) ++failures;
                const clk::time_point g = clk::now();
                if (it >= warmup) {
                    t_view.push_back(us(a, g));
                    bd[0] += us(a, b); bd[1] += us(b, c); bd[2] += us(c, e); bd[3] += us(e, g);
                }
            }
            clc_ctx_destroy(raw);
        }
        for (double& v : bd) v /= frames > 0 ? frames : 1;
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
                "\"detect_us\": %.1f, \"detect_us_p95\": %.1f, \"detect_steps_us\": {\"total_p50\": %.1f, \"c_entry_image_in_to_results_in_pinned_memory\": %.1f, \"keypoints_copy\": %.1f, \"regions_and_features\": %.1f, \"descriptor_copy_publish\": %.1f}, \"match_us\": %.1f, \"match_us_p95\": %.1f, \"pose_us\": %.1f, \"pose_us_p95\": %.1f, "
                "\"frame_us\": %.1f, \"frame_us_p95\": %.1f, \"pair_match_us\": %.1f, \"pair_match_us_p95\": %.1f, \"failures\": %d, "
                "\"same_results_every_frame\": %s}\n",
                frames, warmup, w, h, kp[0], kp[1], data.mapRegions->RegionCount(), n_map[0], n_map[1], n_inl[0], n_inl[1], n_pair,
                pct(t_detect, 0.5), pct(t_detect, 0.95), pct(t_view, 0.5), bd[0], bd[1], bd[2], bd[3], pct(t_match, 0.5), pct(t_match, 0.95), pct(t_pose, 0.5), pct(t_pose, 0.95),
                pct(t_frame, 0.5), pct(t_frame, 0.95), pct(t_pair, 0.5), pct(t_pair, 0.95), failures, failures == 0 ? "true" : "false");
    return failures == 0 ? 0 : 3;
}
