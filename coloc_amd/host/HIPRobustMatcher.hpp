// HIPRobustMatcher.hpp -- the essential-matrix path of coloc::RobustMatcher (reference
// include/coloc/RobustMatcher.hpp:153-186, 372-424) over the C ABI of libcoloc_hip.so.
//
//   bool filterEssential(intrinsics1, intrinsics2, x1, x2, relativePose_info, params, findPose)      :153-186
//        ACRANSAC(ACKernelAdaptorEssential<FivePointSolver, SymmetricEpipolarDistanceError>, inliers, 256, &E,
//        initial_residual_tolerance = +inf)  ->  clc_essential_acransac (a-contrario five-point RANSAC on the GPU);
//        failure iff inliers < 2.5 x 5; with findPose: RelativePoseFromEssential(bearings1, bearings2, E, inliers, &pose)
//   bool RelativePoseFromEssential(...)                                                                :176-183
//   bool computeRelativePose(relativePose, pair, regions, putativeMatches)                            :372-424
//   bool filterMatchesPair(pair, regions, putativeMatches, geometricMatches, relativePoses)           :426-453  (coloc.hpp:296)
//   void filterMatches(regions, putativeMatches, geometricMatches, relativePoses)                     :455-483  (coloc.hpp:167, 412)
//        host arithmetic, restated from OpenMVG's published multiview code (motion_from_essential.hpp): the four
//        (R, t) candidates of E = U diag(1,1,0) V^T (R = U W V^T or U W^T V^T, t = +-u3), every inlier triangulated
//        (DLT on the bearing vectors) under each candidate, the candidate with most points in front of both cameras
//        wins provided the runner-up has < 0.7 of its count.
//   bool matchMaps(scene1, scene2, commonFeatures, poseDiff, rotDiff)                                  :241-370  (coloc.hpp:326, 443)
//        as the reference's live code does: every putative map-to-map match is kept, in order (:346, :359-360); the fundamental
//        matrix predicted from the known displacement, F = Kinv^T rotDiff^T K^T [K rotDiff poseDiff / |poseDiff|]_x (:317-327), is
//        only used to log f1^T F f2 per match to "guidedmatches2.txt" (:334-349).  Host arithmetic, no GPU work.
// x1 / x2 are 2 x N UNDISTORTED pixel coordinates (computeRelativePose undistorts with get_ud_pixel, :391-397).
// Status convention as the reference: EXIT_SUCCESS / EXIT_FAILURE through bool, FALSE MEANS SUCCESS.
// Only model 'E' (the one coloc_node.cpp:87 selects) is provided; 'F' / 'H' stay with OpenMVG on the host.
#pragma once

#include <array>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <vector>

#include "coloc_hip.h"
#include "coloc_hip_geometry.hpp"

namespace coloc {
namespace hipgeom {

// cyclic Jacobi on a symmetric N x N matrix: A = V diag(w) V^T, eigenvalues unsorted
template <int N>
inline void jacobi_eigen(std::array<double, N * N> A, std::array<double, N>& w, std::array<double, N * N>& V)
{
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) V[N * i + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < N; ++p) for (int q = p + 1; q < N; ++q) off += A[N * p + q] * A[N * p + q];
        if (off < 1e-300) break;
        for (int p = 0; p < N; ++p) {
            for (int q = p + 1; q < N; ++q) {
                const double apq = A[N * p + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[N * q + q] - A[N * p + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < N; ++k) {
                    const double akp = A[N * k + p], akq = A[N * k + q];
                    A[N * k + p] = c * akp - s * akq;
                    A[N * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < N; ++k) {
                    const double apk = A[N * p + k], aqk = A[N * q + k];
                    A[N * p + k] = c * apk - s * aqk;
                    A[N * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < N; ++k) {
                    const double vkp = V[N * k + p], vkq = V[N * k + q];
                    V[N * k + p] = c * vkp - s * vkq;
                    V[N * k + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    for (int i = 0; i < N; ++i) w[i] = A[N * i + i];
}

inline openMVG::Mat3 mul(const openMVG::Mat3& A, const openMVG::Mat3& B)
{
    openMVG::Mat3 C;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C(i, j) = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j);
    return C;
}
inline openMVG::Mat3 transpose(const openMVG::Mat3& A)
{
    openMVG::Mat3 T;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T(i, j) = A(j, i);
    return T;
}
inline openMVG::Vec3 cross(const openMVG::Vec3& a, const openMVG::Vec3& b)
{
    return openMVG::Vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

// MotionFromEssential: the four relative poses (R, C = -R^T t) of E; U and V forced to determinant +1
inline void motion_from_essential(const openMVG::Mat3& E, std::vector<openMVG::geometry::Pose3>* relative_poses)
{
    std::array<double, 9> EtE{};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) EtE[3 * i + j] = E(0, i) * E(0, j) + E(1, i) * E(1, j) + E(2, i) * E(2, j);
    std::array<double, 3> w;
    std::array<double, 9> Vq;
    jacobi_eigen<3>(EtE, w, Vq);
    int order[3] = { 0, 1, 2 };
    for (int a = 0; a < 3; ++a) for (int b = a + 1; b < 3; ++b) if (w[order[b]] > w[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
    openMVG::Vec3 v[3], u[3];
    for (int k = 0; k < 2; ++k) {
        v[k] = openMVG::Vec3(Vq[0 + order[k]], Vq[3 + order[k]], Vq[6 + order[k]]);
        const double s = std::sqrt(w[order[k]] > 0 ? w[order[k]] : 0.0);
        for (int i = 0; i < 3; ++i) u[k][i] = (E(i, 0) * v[k][0] + E(i, 1) * v[k][1] + E(i, 2) * v[k][2]) / (s > 0 ? s : 1.0);
    }
    v[2] = cross(v[0], v[1]);                            // det(V) = +1
    u[2] = cross(u[0], u[1]);                            // det(U) = +1
    const double un = std::sqrt(u[2][0] * u[2][0] + u[2][1] * u[2][1] + u[2][2] * u[2][2]);
    for (int i = 0; i < 3; ++i) u[2][i] /= (un > 0 ? un : 1.0);
    // every entry of every matrix is assigned: behind the name Mat3 stands Eigen::Matrix3d in the reference tree, whose default
    // constructor leaves the coefficients uninitialised
    openMVG::Mat3 U, Vt, W;
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { U(i, k) = u[k][i]; Vt(k, i) = v[k][i]; W(i, k) = 0.0; }
    W(0, 1) = -1.0; W(1, 0) = 1.0; W(2, 2) = 1.0;
    const openMVG::Mat3 R[2] = { mul(mul(U, W), Vt), mul(mul(U, transpose(W)), Vt) };
    const openMVG::Vec3 t[2] = { u[2], openMVG::Vec3(-u[2][0], -u[2][1], -u[2][2]) };
    relative_poses->clear();
    for (int i = 0; i < 4; ++i) {
        const openMVG::Mat3& Ri = R[i % 2];
        const openMVG::Vec3& ti = t[i / 2];
        openMVG::Vec3 C;
        for (int a = 0; a < 3; ++a) C[a] = -(Ri(0, a) * ti[0] + Ri(1, a) * ti[1] + Ri(2, a) * ti[2]);
        relative_poses->emplace_back(Ri, C);
    }
}

// TriangulateDLT of two bearing vectors, camera 1 at the origin, camera 2 = pose2; false if the point is at infinity
inline bool triangulate_dlt(const openMVG::Vec3& f1, const openMVG::Vec3& f2, const openMVG::geometry::Pose3& pose2, openMVG::Vec3* X)
{
    double P1[3][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 } }, P2[3][4];
    const openMVG::Vec3 t = pose2.translation();
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P2[i][j] = pose2.rotation()(i, j); P2[i][3] = t[i]; }
    double D[4][4];
    for (int j = 0; j < 4; ++j) {
        D[0][j] = f1[0] * P1[2][j] - f1[2] * P1[0][j];
        D[1][j] = f1[1] * P1[2][j] - f1[2] * P1[1][j];
        D[2][j] = f2[0] * P2[2][j] - f2[2] * P2[0][j];
        D[3][j] = f2[1] * P2[2][j] - f2[2] * P2[1][j];
    }
    std::array<double, 16> DtD{};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) DtD[4 * i + j] += D[k][i] * D[k][j];
    std::array<double, 4> w;
    std::array<double, 16> V;
    jacobi_eigen<4>(DtD, w, V);
    int m = 0;
    for (int i = 1; i < 4; ++i) if (w[i] < w[m]) m = i;
    const double h = V[12 + m];
    if (std::fabs(h) < 1e-300) return false;
    *X = openMVG::Vec3(V[0 + m] / h, V[4 + m] / h, V[8 + m] / h);
    return true;
}

inline bool RelativePoseFromEssential(const openMVG::Mat3X& x1, const openMVG::Mat3X& x2, const openMVG::Mat3& E,
                                      const std::vector<uint32_t>& bearing_vector_index_to_use, openMVG::geometry::Pose3* relative_pose,
                                      std::vector<uint32_t>* vec_selected_points = nullptr, std::vector<openMVG::Vec3>* vec_points = nullptr,
                                      const double positive_depth_solution_ratio = 0.7)
{
    std::vector<openMVG::geometry::Pose3> relative_poses;
    motion_from_essential(E, &relative_poses);
    std::vector<uint32_t> cheirality_accumulator(relative_poses.size(), 0);
    std::vector<std::vector<uint32_t>> vec_newInliers(relative_poses.size());
    std::vector<std::vector<openMVG::Vec3>> vec_3D(relative_poses.size());
    for (size_t i = 0; i < relative_poses.size(); ++i) {
        const openMVG::geometry::Pose3& pose2 = relative_poses[i];
        const openMVG::Vec3 t2 = pose2.translation();
        for (const uint32_t k : bearing_vector_index_to_use) {
            const openMVG::Vec3 f1(x1(0, k), x1(1, k), x1(2, k)), f2(x2(0, k), x2(1, k), x2(2, k));
            openMVG::Vec3 X;
            if (!triangulate_dlt(f1, f2, pose2, &X)) continue;
            // CheiralityTest: the point is in front of both bearing vectors
            openMVG::Vec3 X2;
            for (int a = 0; a < 3; ++a) X2[a] = pose2.rotation()(a, 0) * X[0] + pose2.rotation()(a, 1) * X[1] + pose2.rotation()(a, 2) * X[2] + t2[a];
            const double d1 = f1[0] * X[0] + f1[1] * X[1] + f1[2] * X[2], d2 = f2[0] * X2[0] + f2[1] * X2[1] + f2[2] * X2[2];
            if (d1 > 0.0 && d2 > 0.0) {
                ++cheirality_accumulator[i];
                vec_newInliers[i].push_back(k);
                vec_3D[i].push_back(X);
            }
        }
    }
    size_t index = 0;
    for (size_t i = 1; i < cheirality_accumulator.size(); ++i) if (cheirality_accumulator[i] > cheirality_accumulator[index]) index = i;
    if (cheirality_accumulator[index] == 0) return false;
    *relative_pose = relative_poses[index];
    if (vec_selected_points) *vec_selected_points = vec_newInliers[index];
    if (vec_points) *vec_points = vec_3D[index];
    std::vector<uint32_t> sorted(cheirality_accumulator);
    for (size_t a = 0; a < sorted.size(); ++a) for (size_t b = a + 1; b < sorted.size(); ++b) if (sorted[b] < sorted[a]) { const uint32_t t = sorted[a]; sorted[a] = sorted[b]; sorted[b] = t; }
    const double ratio = sorted[sorted.size() - 2] / static_cast<double>(sorted[sorted.size() - 1]);
    return ratio < positive_depth_solution_ratio;
}

} // namespace hipgeom

typedef std::map<openMVG::Pair, openMVG::sfm::RelativePose_Info> InterPoseMap;     // colocData.hpp:25

class HIPRobustMatcher {
public:
    int iterationCount = 256;                              // RobustMatcher.hpp:34
    uint64_t seed = 1;

    explicit HIPRobustMatcher(colocParams& params) : params(&params)
    {
        const int rc = clc_ctx_create(0, nullptr, nullptr, &ctx_);
        if (rc != CLC_OK) {
            std::cerr << "HIPRobustMatcher: clc_ctx_create failed: " << clc_status_string(rc) << std::endl;
            ctx_ = nullptr;
        }
    }
    HIPRobustMatcher(const HIPRobustMatcher&) = delete;
    HIPRobustMatcher& operator=(const HIPRobustMatcher&) = delete;
    ~HIPRobustMatcher() { if (ctx_) clc_ctx_destroy(ctx_); }

    bool filterEssential(const openMVG::cameras::IntrinsicBase* intrinsics1, const openMVG::cameras::IntrinsicBase* intrinsics2,
                         const openMVG::Mat& x1, const openMVG::Mat& x2, openMVG::sfm::RelativePose_Info& relativePose_info,
                         colocParams& prm, bool findPose)
    {
        if (!intrinsics1 || !intrinsics2 || !ctx_) return EXIT_FAILURE;
        const int n = static_cast<int>(x1.cols());
        std::vector<double> p1(2 * static_cast<size_t>(n)), p2(2 * static_cast<size_t>(n));
        // bearing vectors exactly as the reference asks for them (RobustMatcher.hpp:159): IntrinsicBase::operator()(Mat2X) -> Mat3X
        const openMVG::Mat3X norm2Dpt_1 = (*intrinsics1)(x1), norm2Dpt_2 = (*intrinsics2)(x2);
        for (int i = 0; i < n; ++i) {
            p1[2 * i] = x1(0, i); p1[2 * i + 1] = x1(1, i);
            p2[2 * i] = x2(0, i); p2[2 * i + 1] = x2(1, i);
        }
        double K1[9], K2[9], E[9], F[9], emax = 0.0, nfa = 0.0;
        const openMVG::Mat3 k1 = intrinsics1->K(), k2 = intrinsics2->K();
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { K1[3 * i + j] = k1(i, j); K2[3 * i + j] = k2(i, j); }
        std::vector<int32_t> inl(static_cast<size_t>(n > 0 ? n : 1));
        int n_inl = 0, its = 0;
        const int rc = clc_essential_acransac(ctx_, p1.data(), p2.data(), n, K1, K2, prm.imageSize.first, prm.imageSize.second, iterationCount,
                                              seed++, relativePose_info.initial_residual_tolerance, E, F, nullptr, inl.data(), &n_inl, &emax,
                                              &nfa, &its);
        if (rc != CLC_OK) {
            std::cerr << "HIPRobustMatcher: clc_essential_acransac: " << clc_last_error_string(ctx_) << std::endl;
            return EXIT_FAILURE;
        }
        relativePose_info.vec_inliers.assign(inl.begin(), inl.begin() + n_inl);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) relativePose_info.essential_matrix(i, j) = E[3 * i + j];
        relativePose_info.found_residual_precision = emax;
        if (relativePose_info.vec_inliers.size() < 2.5 * 5) return EXIT_FAILURE;
        if (findPose) {
            openMVG::geometry::Pose3 relative_pose;
            if (!hipgeom::RelativePoseFromEssential(norm2Dpt_1, norm2Dpt_2, relativePose_info.essential_matrix, relativePose_info.vec_inliers,
                                                    &relative_pose))
                return EXIT_FAILURE;
            relativePose_info.relativePose = relative_pose;
        }
        return EXIT_SUCCESS;
    }

    // RobustMatcher::filterFundamental (:128-151) and ::filterHomography (:188-230).  NOT on the GPU path, on purpose and by name: the
    // hot path SURVEY.md section 8 (f-2) scopes is the essential-matrix model, the one `coloc_node.cpp:87` selects ('E'); the other two
    // need OpenMVG's seven-point / four-point solvers with its point normalisation (ACKernelAdaptor) and, for 'H', OpenCV's
    // decomposeHomographyMat (RobustMatcher.hpp:106-126) -- none of it in the reference tree.  They keep the reference's signatures so
    // that code that names them compiles, return EXIT_FAILURE like a failed estimate, and say why: lastStatus() ==
    // kModelNotOnGpuPath (a caller can fall back to the reference's CPU RobustMatcher for these two models).
    enum Status { kOk = 0, kEstimateFailed = 1, kModelNotOnGpuPath = 2 };
    Status lastStatus() const { return last_status_; }
    bool filterFundamental(const openMVG::cameras::IntrinsicBase*, const openMVG::cameras::IntrinsicBase*, const openMVG::Mat&, const openMVG::Mat&,
                           openMVG::sfm::RelativePose_Info&, colocParams&, bool)
    {
        return not_on_gpu_path('F', "RobustMatcher.hpp:128-151 (SevenPointSolver + EpipolarDistanceError)");
    }
    bool filterHomography(const openMVG::cameras::IntrinsicBase*, const openMVG::cameras::IntrinsicBase*, const openMVG::Mat&, const openMVG::Mat&,
                          openMVG::sfm::RelativePose_Info&, colocParams&, bool)
    {
        return not_on_gpu_path('H', "RobustMatcher.hpp:188-230 (FourPointSolver + AsymmetricError, cv::decomposeHomographyMat)");
    }

    // RobustMatcher::computeRelativePose (:372-424): positions of the pair's putative matches, undistorted through each camera's
    // radial-K3 model, into the filter selected by params->model (:399-405).  Only the essential-matrix model ('E', the one the
    // reference's own launch code uses) runs on the GPU; 'F' / 'H' report kModelNotOnGpuPath (above).
    bool computeRelativePose(openMVG::sfm::RelativePose_Info& relativePose, openMVG::Pair current_pair, FeatureMap& regions,
                             openMVG::matching::PairWiseMatches& putativeMatches)
    {
        const uint32_t I = std::min(current_pair.first, current_pair.second);
        const uint32_t J = std::max(current_pair.first, current_pair.second);
        const std::vector<openMVG::matching::IndMatch> pairMatches = putativeMatches[current_pair];
        openMVG::Mat xL(2, pairMatches.size()), xR(2, pairMatches.size());
        const auto& Ka = params->K[current_pair.first];
        const auto& Kb = params->K[current_pair.second];
        const auto& da = params->dist[current_pair.first];
        const auto& db = params->dist[current_pair.second];
        const openMVG::cameras::Pinhole_Intrinsic_Radial_K3 camL(params->imageSize.first, params->imageSize.second, Ka(0, 0), Ka(0, 2), Ka(1, 2), da[0],
                                                                 da[1], da[2]),
            camR(params->imageSize.first, params->imageSize.second, Kb(0, 0), Kb(0, 2), Kb(1, 2), db[0], db[1], db[2]);
        for (size_t k = 0; k < pairMatches.size(); ++k) {
            const auto pi = regions.at(I)->GetRegionPosition(pairMatches[k].i_);
            const auto pj = regions.at(J)->GetRegionPosition(pairMatches[k].j_);
            const openMVG::Vec2 ui = camL.get_ud_pixel(openMVG::Vec2(pi[0], pi[1])), uj = camR.get_ud_pixel(openMVG::Vec2(pj[0], pj[1]));
            xL(0, k) = ui[0]; xL(1, k) = ui[1];
            xR(0, k) = uj[0]; xR(1, k) = uj[1];
        }
        bool status = EXIT_FAILURE;
        last_status_ = kOk;
        if (params->model == 'E') {
            status = filterEssential(&camL, &camR, xL, xR, relativePose, *params, true);
            if (status == EXIT_FAILURE) last_status_ = kEstimateFailed;
        }
        else if (params->model == 'F') status = filterFundamental(&camL, &camR, xL, xR, relativePose, *params, true);
        else if (params->model == 'H') status = filterHomography(&camL, &camR, xL, xR, relativePose, *params, true);
        else status = not_on_gpu_path(params->model, "RobustMatcher.hpp:399-405 knows 'E', 'F', 'H'");
        if (status == EXIT_FAILURE && last_status_ != kModelNotOnGpuPath) std::cerr << "Unable to estimate relative pose." << std::endl;
        return status;
    }

    // RobustMatcher::filterMatchesPair (:426-453): geometricMatches[pair] = the putative matches the model keeps (none -> entry left
    // alone), relativePoses[pair] = the model -- also when the estimate failed, like the reference
    bool filterMatchesPair(openMVG::Pair currentPair, FeatureMap& regions, openMVG::matching::PairWiseMatches& putativeMatches,
                           openMVG::matching::PairWiseMatches& geometricMatches, InterPoseMap& relativePoses)
    {
        const std::vector<openMVG::matching::IndMatch> pairMatches = putativeMatches[currentPair];
        openMVG::sfm::RelativePose_Info relativePose;
        const bool status = computeRelativePose(relativePose, currentPair, regions, putativeMatches);
        std::vector<openMVG::matching::IndMatch> kept;
        for (size_t ic = 0; ic < relativePose.vec_inliers.size(); ++ic) kept.push_back(pairMatches[relativePose.vec_inliers[ic]]);
        if (!kept.empty()) geometricMatches[currentPair] = kept;
        relativePoses[currentPair] = relativePose;
        return status;
    }

    // RobustMatcher::filterMatches (:455-483): every pair that has putative matches
    void filterMatches(FeatureMap& regions, openMVG::matching::PairWiseMatches& putativeMatches, openMVG::matching::PairWiseMatches& geometricMatches,
                       InterPoseMap& relativePoses)
    {
        std::vector<openMVG::Pair> pairs;
        for (const auto& kv : putativeMatches) pairs.push_back(kv.first);
        for (const openMVG::Pair& pr : pairs) (void)filterMatchesPair(pr, regions, putativeMatches, geometricMatches, relativePoses);
    }

    // RobustMatcher::matchMaps (:241-370), called straight after matcher.matchMapFeatures (coloc.hpp:323-326, :439-443): the live
    // code keeps EVERY putative match (the filterHomography variant above it is commented out in the reference) and writes the
    // epipolar residual of each under the fundamental matrix the known displacement predicts to guidedmatches2.txt, one line
    // "res,xL0,xL1,xR0,xR1" per match (:347).  poseDiff / rotDiff are taken by const reference: coloc.hpp:443 hands over the
    // result of a call.  Returns EXIT_SUCCESS (false).
    bool matchMaps(std::unique_ptr<openMVG::features::AKAZE_Binary_Regions>& scene1, std::unique_ptr<openMVG::features::AKAZE_Binary_Regions>& scene2,
                   std::vector<openMVG::matching::IndMatch>& commonFeatures, const openMVG::Vec3& poseDiff, const openMVG::Mat3& rotDiff)
    {
        const std::vector<openMVG::matching::IndMatch> putativeMatches = commonFeatures;
        commonFeatures.clear();
        const openMVG::features::PointFeatures featI = scene1->GetRegionsPositions();
        const openMVG::features::PointFeatures featJ = scene2->GetRegionsPositions();
        openMVG::Mat xL(2, putativeMatches.size()), xR(2, putativeMatches.size());
        for (size_t k = 0; k < putativeMatches.size(); ++k) {
            xL(0, k) = featI[putativeMatches[k].i_].x(); xL(1, k) = featI[putativeMatches[k].i_].y();
            xR(0, k) = featJ[putativeMatches[k].j_].x(); xR(1, k) = featJ[putativeMatches[k].j_].y();
        }
        // both cameras from params->K[0] / dist[0], as written at :307-311
        const auto& K0 = params->K[0];
        const auto& d0 = params->dist[0];
        const openMVG::cameras::Pinhole_Intrinsic_Radial_K3 camL(params->imageSize.first, params->imageSize.second, K0(0, 0), K0(0, 2), K0(1, 2), d0[0],
                                                                 d0[1], d0[2]),
            camR(params->imageSize.first, params->imageSize.second, K0(0, 0), K0(0, 2), K0(1, 2), d0[0], d0[1], d0[2]);
        // A = K rotDiff (poseDiff / |poseDiff|), C = [A]_x, F = Kinv^T rotDiff^T K^T C (:315-325)
        const double nrm = std::sqrt(poseDiff[0] * poseDiff[0] + poseDiff[1] * poseDiff[1] + poseDiff[2] * poseDiff[2]);
        const openMVG::Mat3 KR = hipgeom::mul(camL.K(), rotDiff);
        double A[3];
        for (int i = 0; i < 3; ++i) A[i] = KR(i, 0) * (poseDiff[0] / nrm) + KR(i, 1) * (poseDiff[1] / nrm) + KR(i, 2) * (poseDiff[2] / nrm);
        openMVG::Mat3 C;
        C(0, 0) = 0.0;   C(0, 1) = -A[2]; C(0, 2) = A[1];
        C(1, 0) = A[2];  C(1, 1) = 0.0;   C(1, 2) = -A[0];
        C(2, 0) = -A[1]; C(2, 1) = A[0];  C(2, 2) = 0.0;
        const openMVG::Mat3 F = hipgeom::mul(hipgeom::mul(hipgeom::mul(hipgeom::transpose(camR.Kinv()), hipgeom::transpose(rotDiff)),
                                                          hipgeom::transpose(camL.K())), C);
        std::ofstream myfile;
        myfile.open("guidedmatches2.txt");
        for (size_t k = 0; k < putativeMatches.size(); ++k) {
            const double f1[3] = { xL(0, k), xL(1, k), 1.0 }, f2[3] = { xR(0, k), xR(1, k), 1.0 };
            double res = 0.0;
            for (int i = 0; i < 3; ++i) res += f1[i] * (F(i, 0) * f2[0] + F(i, 1) * f2[1] + F(i, 2) * f2[2]);
            commonFeatures.push_back(putativeMatches[k]);
            myfile << res << "," << xL(0, k) << "," << xL(1, k) << "," << xR(0, k) << "," << xR(1, k) << std::endl;
        }
        myfile.close();
        return EXIT_SUCCESS;
    }

private:
    bool not_on_gpu_path(const char model, const char* where)
    {
        last_status_ = kModelNotOnGpuPath;
        std::cerr << "HIPRobustMatcher: filtering model '" << model << "' is not on the GPU path (" << where
                  << "); only the essential-matrix model 'E' is -- use the reference's CPU RobustMatcher for this model." << std::endl;
        return EXIT_FAILURE;
    }
    Status last_status_ = kOk;
    clc_ctx* ctx_ = nullptr;
    colocParams* params;
};

} // namespace coloc
