// localizer_driver.cpp -- exercises HIPLocalizer / HIPRobustMatcher the way ColoC drives Localizer / RobustMatcher
// (reference include/coloc/coloc.hpp:219-225 intraPoseEstimator, :296 filterMatchesPair -> computeRelativePose ->
// filterEssential) and dumps the results as raw doubles for tests/test_gpu_localizer.py.
// usage: localizer_driver <dir>    reads <dir>/loc.bin, <dir>/twoview.bin; writes <dir>/loc_out.bin, <dir>/twoview_out.bin
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "HIPLocalizer.hpp"
#include "HIPRobustMatcher.hpp"

using namespace openMVG;

static std::vector<double> slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    std::vector<double> v(static_cast<size_t>(f.tellg()) / 8);
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(v.size() * 8));
    return v;
}
static void dump(const std::string& path, const std::vector<double>& v)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), static_cast<std::streamsize>(v.size() * 8));
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s dir\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    // ---- loc.bin: [w, h, f, ppx, ppy, k1, k2, k3, n_map, n_feat, n_match, map X (3 n_map), feature xy (2 n_feat), matches (2 n_match: map idx, feat idx)]
    {
        const std::vector<double> in = slurp(dir + "/loc.bin");
        const int w = (int)in[0], h = (int)in[1];
        Mat3 K; K(0, 0) = in[2]; K(1, 1) = in[2]; K(0, 2) = in[3]; K(1, 2) = in[4]; K(2, 2) = 1.0;
        const Vec3 dist(in[5], in[6], in[7]);
        const size_t n_map = (size_t)in[8], n_feat = (size_t)in[9], n_match = (size_t)in[10];
        const double* p = in.data() + 11;
        coloc::colocParams params({ K }, { dist }, 'E', { (size_t)w, (size_t)h }, ".", coloc::DetectorOptions{}, coloc::MatcherOptions{});
        coloc::colocData data;
        for (size_t i = 0; i < n_map; ++i) {
            data.scene.structure[(IndexT)(1000 + i)].X = Vec3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);     // landmark ids are not row numbers
            data.mapRegionIdx.push_back((IndexT)(1000 + i));
        }
        p += 3 * n_map;
        data.regions[0].reset(new features::AKAZE_Binary_Regions);
        for (size_t i = 0; i < n_feat; ++i) data.regions[0]->Features().emplace_back((float)p[2 * i], (float)p[2 * i + 1], 7.0f, 0.0f);
        p += 2 * n_feat;
        matching::IndMatches tracked;
        for (size_t i = 0; i < n_match; ++i) tracked.emplace_back((IndexT)p[2 * i], (IndexT)p[2 * i + 1]);
        coloc::HIPLocalizer localizer(params);
        int idx = 0;
        geometry::Pose3 pose;
        coloc::Cov6 cov;
        float rmse = -1.0f;
        std::vector<uint32_t> inliers;
        const bool status = localizer.localizeImage(idx, pose, data, cov, rmse, tracked, inliers);   // false = success
        std::vector<double> out;
        out.push_back(status ? 1.0 : 0.0);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(pose.rotation()(i, j));
        for (int i = 0; i < 3; ++i) out.push_back(pose.center()[i]);
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) out.push_back(cov(i, j));
        out.push_back(rmse);
        out.push_back((double)inliers.size());
        for (uint32_t v : inliers) out.push_back(v);
        // the second call site of the same solve (Reconstructor::resectionCamera): SfM_Localizer::Localize called directly on
        // prepared Image_Localizer_Match_Data; same sampler seed as the member call above -> the same inlier set
        {
            cameras::Pinhole_Intrinsic_Radial_K3 cam((int)w, (int)h, K(0, 0), K(0, 2), K(1, 2), dist[0], dist[1], dist[2]);
            sfm::Image_Localizer_Match_Data rd;
            rd.error_max = std::numeric_limits<double>::infinity();
            rd.max_iteration = 256;
            localizer.setupTracks(&cam, data, *data.regions.at(0), tracked, &rd);
            geometry::Pose3 pose2;
            const bool ok = coloc::HIP_SfM_Localizer::Localize(coloc::resection::SolverType::P3P_KE_CVPR17, { (size_t)w, (size_t)h }, &cam, rd, pose2, 1);
            out.push_back(ok ? 1.0 : 0.0);
            out.push_back((double)rd.vec_inliers.size());
            for (int i = 0; i < 3; ++i) out.push_back(pose2.center()[i]);
            const bool none = coloc::HIP_SfM_Localizer::Localize(coloc::resection::SolverType::P3P_KE_CVPR17, { (size_t)w, (size_t)h }, nullptr, rd, pose2, 1);
            out.push_back(none ? 1.0 : 0.0);
        }
        // localizeImages (config[2]'s batched pose): two cameras' worth of the same data in one call on a FRESH localizer must give exactly
        // what two localizeImage calls in a row give on another fresh one (same sampler seeds in the same order)
        {
            coloc::HIPLocalizer seq(params), bat(params);
            geometry::Pose3 p1, p2;
            coloc::Cov6 c1, c2;
            float r1 = -1.0f, r2 = -1.0f;
            std::vector<uint32_t> i1, i2;
            int id = 0;
            const bool s1 = seq.localizeImage(id, p1, data, c1, r1, tracked, i1);
            const bool s2 = seq.localizeImage(id, p2, data, c2, r2, tracked, i2);
            std::vector<geometry::Pose3> bp;
            std::vector<coloc::Cov6> bc;
            std::vector<float> br;
            std::vector<matching::IndMatches> bt = { tracked, tracked };
            std::vector<std::vector<uint32_t>> bi;
            const std::vector<bool> bs = bat.localizeImages({ 0, 0 }, bp, data, bc, br, bt, bi);
            bool same = bs.size() == 2 && bs[0] == s1 && bs[1] == s2 && bi[0] == i1 && bi[1] == i2 && br[0] == r1 && br[1] == r2;
            for (int i = 0; same && i < 3; ++i) {
                same = same && bp[0].center()[i] == p1.center()[i] && bp[1].center()[i] == p2.center()[i];
                for (int j = 0; j < 3; ++j) same = same && bp[0].rotation()(i, j) == p1.rotation()(i, j) && bp[1].rotation()(i, j) == p2.rotation()(i, j);
            }
            for (int i = 0; same && i < 6; ++i) for (int j = 0; j < 6; ++j) same = same && bc[0](i, j) == c1(i, j) && bc[1](i, j) == c2(i, j);
            out.push_back(same ? 1.0 : 0.0);
            out.push_back((s1 || s2) ? 1.0 : 0.0);
        }
        dump(dir + "/loc_out.bin", out);
    }
    // ---- twoview.bin: [w, h, f, ppx, ppy, n, x1 (2 n), x2 (2 n)]  (undistorted pixels)
    {
        const std::vector<double> in = slurp(dir + "/twoview.bin");
        const int w = (int)in[0], h = (int)in[1];
        const size_t n = (size_t)in[5];
        Mat3 K; K(0, 0) = in[2]; K(1, 1) = in[2]; K(0, 2) = in[3]; K(1, 2) = in[4]; K(2, 2) = 1.0;
        coloc::colocParams params({ K, K }, { Vec3(0, 0, 0), Vec3(0, 0, 0) }, 'E', { (size_t)w, (size_t)h }, ".", coloc::DetectorOptions{},
                                  coloc::MatcherOptions{});
        const cameras::Pinhole_Intrinsic_Radial_K3 camL(w, h, in[2], in[3], in[4], 0, 0, 0), camR(w, h, in[2], in[3], in[4], 0, 0, 0);
        Mat xL(2, n), xR(2, n);
        for (size_t i = 0; i < n; ++i) {
            xL(0, i) = in[6 + 2 * i]; xL(1, i) = in[6 + 2 * i + 1];
            xR(0, i) = in[6 + 2 * n + 2 * i]; xR(1, i) = in[6 + 2 * n + 2 * i + 1];
        }
        coloc::HIPRobustMatcher robust(params);
        sfm::RelativePose_Info info;
        const bool status = robust.filterEssential(&camL, &camR, xL, xR, info, params, true);
        std::vector<double> out;
        out.push_back(status ? 1.0 : 0.0);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(info.essential_matrix(i, j));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(info.relativePose.rotation()(i, j));
        for (int i = 0; i < 3; ++i) out.push_back(info.relativePose.center()[i]);
        out.push_back(info.found_residual_precision);
        out.push_back((double)info.vec_inliers.size());
        for (uint32_t v : info.vec_inliers) out.push_back(v);
        // the members ColoC actually calls (coloc.hpp:167, 296): filterMatches over regions + putative matches.  Pair (0, 1):
        // feature k of view 0 <-> feature n-1-k of view 1, so the match indices are not the identity
        {
            coloc::FeatureMap regions;
            regions[0].reset(new features::AKAZE_Binary_Regions);
            regions[1].reset(new features::AKAZE_Binary_Regions);
            for (size_t i = 0; i < n; ++i) regions[0]->Features().emplace_back((float)xL(0, i), (float)xL(1, i), 7.0f, 0.0f);
            for (size_t i = 0; i < n; ++i) regions[1]->Features().emplace_back((float)xR(0, n - 1 - i), (float)xR(1, n - 1 - i), 7.0f, 0.0f);
            matching::PairWiseMatches putative, geometric;
            for (size_t i = 0; i < n; ++i) putative[{ 0, 1 }].emplace_back((IndexT)i, (IndexT)(n - 1 - i));
            coloc::InterPoseMap poses;
            coloc::HIPRobustMatcher robust2(params);              // fresh object: same sampler seed as `robust` had
            robust2.filterMatches(regions, putative, geometric, poses);
            const auto& g = geometric[{ 0, 1 }];
            out.push_back((double)g.size());
            out.push_back((double)poses.count({ 0, 1 }));
            size_t consistent = 0;
            for (const auto& m : g) consistent += (m.i_ + m.j_ == n - 1) ? 1 : 0;
            out.push_back((double)consistent);
            for (int i = 0; i < 3; ++i) out.push_back(poses[{ 0, 1 }].relativePose.center()[i]);
            // an unknown model letter ("Unknown filtering type", RobustMatcher.hpp:406-408) says so through a status of its own instead of
            // failing like a bad estimate; 'E', 'F' and 'H' all give an estimate on this scene (tests/test_gpu_robust_models.py checks 'F' / 'H')
            const char keep = params.model;
            double named = 1.0;
            {
                params.model = 'Q';
                coloc::HIPRobustMatcher robust3(params);
                sfm::RelativePose_Info info3;
                const bool st = robust3.computeRelativePose(info3, { 0, 1 }, regions, putative);
                named *= (st == EXIT_FAILURE && robust3.lastStatus() == coloc::HIPRobustMatcher::kModelNotOnGpuPath) ? 1.0 : 0.0;
            }
            for (const char mdl : { 'E', 'F' }) {
                params.model = mdl;
                coloc::HIPRobustMatcher robust4(params);
                sfm::RelativePose_Info info4;
                const bool st = robust4.computeRelativePose(info4, { 0, 1 }, regions, putative);
                named *= (st == EXIT_SUCCESS && robust4.lastStatus() == coloc::HIPRobustMatcher::kOk) ? 1.0 : 0.0;
            }
            params.model = keep;
            out.push_back(named);
        }
        dump(dir + "/twoview_out.bin", out);
    }
    return 0;
}
