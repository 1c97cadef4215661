// robust_models_driver.cpp -- HIPRobustMatcher under the models 'F' and 'H' the way RobustMatcher::computeRelativePose reaches them
// (reference include/coloc/RobustMatcher.hpp:399-405 -> filterFundamental :128-151, filterHomography :188-239 -> decomposeHomography
// :106-126, performChiralityTest :39-104); dumps raw doubles for tests/test_gpu_robust_models.py.
// usage: robust_models_driver <dir>   reads <dir>/general.bin, <dir>/planar.bin: [w, h, f, ppx, ppy, n, x1 (2 n), x2 (2 n)] (undistorted pixels)
//                                     writes <dir>/models_out.bin
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

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

static void put(std::vector<double>& out, const bool status, const sfm::RelativePose_Info& info)
{
    out.push_back(status ? 1.0 : 0.0);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(info.essential_matrix(i, j));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(info.relativePose.rotation()(i, j));
    for (int i = 0; i < 3; ++i) out.push_back(info.relativePose.center()[i]);
    out.push_back(info.found_residual_precision);
    out.push_back((double)info.vec_inliers.size());
    for (uint32_t v : info.vec_inliers) out.push_back(v);
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s dir\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    std::vector<double> out;
    for (const char* name : { "general", "planar" }) {
        const std::vector<double> in = slurp(dir + "/" + name + ".bin");
        const int w = (int)in[0], h = (int)in[1];
        const size_t n = (size_t)in[5];
        Mat3 K; K(0, 0) = in[2]; K(1, 1) = in[2]; K(0, 2) = in[3]; K(1, 2) = in[4]; K(2, 2) = 1.0;
        const char model = name[0] == 'g' ? 'F' : 'H';
        coloc::colocParams params({ K, K }, { Vec3(0, 0, 0), Vec3(0, 0, 0) }, model, { (size_t)w, (size_t)h }, ".", coloc::DetectorOptions{},
                                  coloc::MatcherOptions{});
        const cameras::Pinhole_Intrinsic_Radial_K3 camL(w, h, in[2], in[3], in[4], 0, 0, 0), camR(w, h, in[2], in[3], in[4], 0, 0, 0);
        Mat xL(2, n), xR(2, n);
        for (size_t i = 0; i < n; ++i) {
            xL(0, i) = in[6 + 2 * i]; xL(1, i) = in[6 + 2 * i + 1];
            xR(0, i) = in[6 + 2 * n + 2 * i]; xR(1, i) = in[6 + 2 * n + 2 * i + 1];
        }
        // the filter member itself (seed 1, as the Python binding's default)
        {
            coloc::HIPRobustMatcher robust(params);
            sfm::RelativePose_Info info;
            const bool status = model == 'F' ? robust.filterFundamental(&camL, &camR, xL, xR, info, params, true)
                                             : robust.filterHomography(&camL, &camR, xL, xR, info, params, true);
            put(out, status, info);
            if (model == 'H') {
                // the candidates the vote above chose among: the motions of the ESTIMATED homography
                std::vector<geometry::Pose3> cand;
                robust.decomposeHomography(info.essential_matrix, cand);
                out.push_back((double)cand.size());
                for (const geometry::Pose3& m : cand) {
                    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(m.rotation()(i, j));
                    for (int i = 0; i < 3; ++i) out.push_back(m.center()[i]);
                }
            }
        }
        // the members ColoC calls: filterMatches over regions + putative matches (feature k of view 0 <-> feature n-1-k of view 1)
        {
            coloc::FeatureMap regions;
            regions[0].reset(new features::AKAZE_Binary_Regions);
            regions[1].reset(new features::AKAZE_Binary_Regions);
            for (size_t i = 0; i < n; ++i) regions[0]->Features().emplace_back((float)xL(0, i), (float)xL(1, i), 7.0f, 0.0f);
            for (size_t i = 0; i < n; ++i) regions[1]->Features().emplace_back((float)xR(0, n - 1 - i), (float)xR(1, n - 1 - i), 7.0f, 0.0f);
            matching::PairWiseMatches putative, geometric;
            for (size_t i = 0; i < n; ++i) putative[{ 0, 1 }].emplace_back((IndexT)i, (IndexT)(n - 1 - i));
            coloc::InterPoseMap poses;
            coloc::HIPRobustMatcher robust2(params);
            robust2.filterMatches(regions, putative, geometric, poses);
            const auto& g = geometric[{ 0, 1 }];
            size_t consistent = 0;
            for (const auto& m : g) consistent += (m.i_ + m.j_ == n - 1) ? 1 : 0;
            out.push_back((double)g.size());
            out.push_back((double)consistent);
            out.push_back((double)poses.count({ 0, 1 }));
            out.push_back(robust2.lastStatus() == coloc::HIPRobustMatcher::kOk ? 1.0 : 0.0);
        }
        // too few matches: the estimate fails (fewer than 2.5 x the sample size inliers), with the status of a failed estimate
        {
            const size_t few = model == 'F' ? 12 : 8;
            Mat aL(2, few), aR(2, few);
            for (size_t i = 0; i < few; ++i) { aL(0, i) = xL(0, i); aL(1, i) = xL(1, i); aR(0, i) = xR(0, i); aR(1, i) = xR(1, i); }
            coloc::HIPRobustMatcher robust3(params);
            sfm::RelativePose_Info info;
            const bool status = model == 'F' ? robust3.filterFundamental(&camL, &camR, aL, aR, info, params, true)
                                             : robust3.filterHomography(&camL, &camR, aL, aR, info, params, true);
            out.push_back(status ? 1.0 : 0.0);
            out.push_back((double)info.vec_inliers.size());
        }
        // decomposeHomography on the scene's exact homography (appended to planar.bin after the points): 4 motions {R, t / |t| in the centre slot}
        if (model == 'H') {
            Mat3 H;
            const double* hp = in.data() + 6 + 4 * n;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H(i, j) = hp[3 * i + j];
            coloc::HIPRobustMatcher robust4(params);
            std::vector<geometry::Pose3> motions;
            const bool st = robust4.decomposeHomography(H, motions);
            out.push_back(st ? 1.0 : 0.0);
            out.push_back((double)motions.size());
            for (const geometry::Pose3& m : motions) {
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(m.rotation()(i, j));
                for (int i = 0; i < 3; ++i) out.push_back(m.center()[i]);
            }
            // a pure rotation: one motion, no translation
            Mat3 Hr;
            const double* rp = hp + 9;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Hr(i, j) = rp[3 * i + j];
            std::vector<geometry::Pose3> rot;
            robust4.decomposeHomography(Hr, rot);
            out.push_back((double)rot.size());
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out.push_back(rot.empty() ? 0.0 : rot[0].rotation()(i, j));
            for (int i = 0; i < 3; ++i) out.push_back(rot.empty() ? 0.0 : rot[0].center()[i]);
        }
    }
    std::ofstream f(dir + "/models_out.bin", std::ios::binary);
    f.write(reinterpret_cast<const char*>(out.data()), static_cast<std::streamsize>(out.size() * 8));
    return 0;
}
