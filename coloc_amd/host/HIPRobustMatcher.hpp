// HIPRobustMatcher.hpp -- coloc::RobustMatcher (reference include/coloc/RobustMatcher.hpp) over the C ABI of libcoloc_hip.so: the
// a-contrario filter under all three of its models, the poses that come out of them, and the members ColoC calls.
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
//   bool filterFundamental(...)  :128-151   ('F': seven points, distance to the epipolar line)     -> clc_two_view_acransac
//   bool filterHomography(...)   :188-239   ('H': four points, transfer error; decomposeHomography :106-126 -- cv::decomposeHomographyMat
//        restated on the host -- and performChiralityTest :39-104 for the pose)                     -> clc_two_view_acransac
// coloc_node.cpp:87 selects 'E'; 'F' / 'H' are reachable through colocParams::model like in the reference (:399-405).
#pragma once

#include <array>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <vector>

#include "coloc_hip.h"
#include "coloc_hip_geometry.hpp"

static_assert(CLC_ABI_VERSION >= 4, "this policy header uses entry points of ABI version 4 (clc_two_view_acransac)");

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

// The chirality vote over candidate relative poses: every listed correspondence is triangulated (DLT on the bearing vectors) under each
// candidate, the candidate with most points in front of both cameras wins (the first of equals), and the result is trusted when the
// runner-up has less than `positive_depth_solution_ratio` of its count.  OpenMVG's RelativePoseFromEssential ends in it and
// RobustMatcher::performChiralityTest (RobustMatcher.hpp:39-104) is the same code over the candidates of a homography.
inline bool chirality_vote(const openMVG::Mat3X& x1, const openMVG::Mat3X& x2, const std::vector<uint32_t>& bearing_vector_index_to_use,
                           const std::vector<openMVG::geometry::Pose3>& relative_poses, openMVG::geometry::Pose3* relative_pose,
                           std::vector<uint32_t>* vec_selected_points, std::vector<openMVG::Vec3>* vec_points,
                           const double positive_depth_solution_ratio)
{
    if (relative_poses.empty()) return false;
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
    if (relative_pose) *relative_pose = relative_poses[index];
    if (vec_selected_points) *vec_selected_points = vec_newInliers[index];
    if (vec_points) *vec_points = vec_3D[index];
    std::vector<uint32_t> sorted(cheirality_accumulator);
    for (size_t a = 0; a < sorted.size(); ++a) for (size_t b = a + 1; b < sorted.size(); ++b) if (sorted[b] < sorted[a]) { const uint32_t t = sorted[a]; sorted[a] = sorted[b]; sorted[b] = t; }
    if (sorted.size() < 2) return true;                  // (one candidate: nothing to compare with; the reference indexes rbegin()[1] here)
    const double ratio = sorted[sorted.size() - 2] / static_cast<double>(sorted[sorted.size() - 1]);
    return ratio < positive_depth_solution_ratio;
}

inline bool RelativePoseFromEssential(const openMVG::Mat3X& x1, const openMVG::Mat3X& x2, const openMVG::Mat3& E,
                                      const std::vector<uint32_t>& bearing_vector_index_to_use, openMVG::geometry::Pose3* relative_pose,
                                      std::vector<uint32_t>* vec_selected_points = nullptr, std::vector<openMVG::Vec3>* vec_points = nullptr,
                                      const double positive_depth_solution_ratio = 0.7)
{
    std::vector<openMVG::geometry::Pose3> relative_poses;
    motion_from_essential(E, &relative_poses);
    return chirality_vote(x1, x2, bearing_vector_index_to_use, relative_poses, relative_pose, vec_selected_points, vec_points,
                          positive_depth_solution_ratio);
}

// cv::decomposeHomographyMat(H, K, rotations, translations, normals) as RobustMatcher::decomposeHomography calls it (RobustMatcher.hpp:
// 106-126).  OpenCV is not part of the reference tree; this restates the analytical method its implementation documents (E. Malis,
// M. Vargas, "Deeper understanding of the homography decomposition for vision-based control", INRIA RR-6303, 2007):
//   Hn = K^-1 H K scaled by 1 / (its middle singular value);  S = Hn^T Hn - I;  a pure rotation (|S|_inf < 1e-3) gives the one motion
//   {Hn, 0, 0};  otherwise the two plane normals n_a, n_b come from the minors of S around its largest diagonal entry, the translations
//   (in the first frame) from the norms r^2 = 2 + tr S + v, |t|^2 = 2 + tr S - v, v = 2 sqrt(1 + tr S - M00 - M11 - M22), and
//   R = Hn (I - (2 / v) t* n^T), negated when its determinant is negative;  four motions {Ra, ta, na}, {Ra, -ta, -na}, {Rb, tb, nb},
//   {Rb, -tb, -nb} with t = R t*.
struct HomographyMotion { openMVG::Mat3 R; openMVG::Vec3 t, n; };
inline double opposite_of_minor(const openMVG::Mat3& M, const int row, const int col)
{
    const int x1 = col == 0 ? 1 : 0, x2 = col == 2 ? 1 : 2, y1 = row == 0 ? 1 : 0, y2 = row == 2 ? 1 : 2;
    return M(y1, x2) * M(y2, x1) - M(y1, x1) * M(y2, x2);
}
inline double det3(const openMVG::Mat3& A)
{
    return A(0, 0) * (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) - A(0, 1) * (A(1, 0) * A(2, 2) - A(1, 2) * A(2, 0)) +
           A(0, 2) * (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0));
}
inline int decompose_homography_mat(const openMVG::Mat3& H, const openMVG::Mat3& K, std::vector<HomographyMotion>* motions)
{
    motions->clear();
    // K^-1 of an upper-triangular calibration matrix
    const double fx = K(0, 0), sk = K(0, 1), cx = K(0, 2), fy = K(1, 1), cy = K(1, 2);
    openMVG::Mat3 Ki;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ki(i, j) = 0.0;
    Ki(0, 0) = 1.0 / fx; Ki(0, 1) = -sk / (fx * fy); Ki(0, 2) = (sk * cy - cx * fy) / (fx * fy);
    Ki(1, 1) = 1.0 / fy; Ki(1, 2) = -cy / fy; Ki(2, 2) = 1.0;
    openMVG::Mat3 Hn = mul(mul(Ki, H), K);
    // removeScale: divide by the middle singular value (square roots of the eigenvalues of Hn^T Hn)
    {
        std::array<double, 9> HtH{};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) HtH[3 * i + j] = Hn(0, i) * Hn(0, j) + Hn(1, i) * Hn(1, j) + Hn(2, i) * Hn(2, j);
        std::array<double, 3> w;
        std::array<double, 9> V;
        jacobi_eigen<3>(HtH, w, V);
        double a = w[0], b = w[1], c = w[2], t;
        if (a < b) { t = a; a = b; b = t; }
        if (b < c) { t = b; b = c; c = t; if (a < b) { t = a; a = b; b = t; } }
        const double mid = std::sqrt(b > 0.0 ? b : 0.0);
        if (!(mid > 0.0)) return 0;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Hn(i, j) /= mid;
    }
    openMVG::Mat3 S = mul(transpose(Hn), Hn);
    S(0, 0) -= 1.0; S(1, 1) -= 1.0; S(2, 2) -= 1.0;
    double ninf = 0.0;                                   // cv::norm(S, NORM_INF): the largest absolute entry
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) ninf = std::max(ninf, std::fabs(S(i, j)));
    if (ninf < 0.001) {
        HomographyMotion m;
        m.R = Hn; m.t = openMVG::Vec3(0, 0, 0); m.n = openMVG::Vec3(0, 0, 0);
        motions->push_back(m);
        return 1;
    }
    auto signd = [](const double x) { return x >= 0.0 ? 1 : -1; };
    const double M00 = opposite_of_minor(S, 0, 0), M11 = opposite_of_minor(S, 1, 1), M22 = opposite_of_minor(S, 2, 2);
    const double rtM00 = std::sqrt(M00), rtM11 = std::sqrt(M11), rtM22 = std::sqrt(M22);
    const double M01 = opposite_of_minor(S, 0, 1), M12 = opposite_of_minor(S, 1, 2), M02 = opposite_of_minor(S, 0, 2);
    const int e12 = signd(M12), e02 = signd(M02), e01 = signd(M01);
    const double nS00 = std::fabs(S(0, 0)), nS11 = std::fabs(S(1, 1)), nS22 = std::fabs(S(2, 2));
    int indx = 0;
    if (nS00 < nS11) { indx = 1; if (nS11 < nS22) indx = 2; }
    else if (nS00 < nS22) indx = 2;
    openMVG::Vec3 npa, npb;
    switch (indx) {
    case 0:
        npa[0] = S(0, 0);               npb[0] = S(0, 0);
        npa[1] = S(0, 1) + rtM22;       npb[1] = S(0, 1) - rtM22;
        npa[2] = S(0, 2) + e12 * rtM11; npb[2] = S(0, 2) - e12 * rtM11;
        break;
    case 1:
        npa[0] = S(0, 1) + rtM22;       npb[0] = S(0, 1) - rtM22;
        npa[1] = S(1, 1);               npb[1] = S(1, 1);
        npa[2] = S(1, 2) - e02 * rtM00; npb[2] = S(1, 2) + e02 * rtM00;
        break;
    default:
        npa[0] = S(0, 2) + e01 * rtM11; npb[0] = S(0, 2) - e01 * rtM11;
        npa[1] = S(1, 2) + rtM00;       npb[1] = S(1, 2) - rtM00;
        npa[2] = S(2, 2);               npb[2] = S(2, 2);
        break;
    }
    const double traceS = S(0, 0) + S(1, 1) + S(2, 2);
    const double v = 2.0 * std::sqrt(1.0 + traceS - M00 - M11 - M22);
    const double ESii = signd(S(indx, indx));
    const double r_2 = 2.0 + traceS + v, nt_2 = 2.0 + traceS - v;
    const double r = std::sqrt(r_2), n_t = std::sqrt(nt_2);
    auto norm3 = [](const openMVG::Vec3& a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
    const double na_n = norm3(npa), nb_n = norm3(npb);
    openMVG::Vec3 na, nb, ta_star, tb_star;
    for (int i = 0; i < 3; ++i) { na[i] = npa[i] / na_n; nb[i] = npb[i] / nb_n; }
    const double half_nt = 0.5 * n_t, esii_t_r = ESii * r;
    for (int i = 0; i < 3; ++i) {
        ta_star[i] = half_nt * (esii_t_r * nb[i] - n_t * na[i]);
        tb_star[i] = half_nt * (esii_t_r * na[i] - n_t * nb[i]);
    }
    auto rotation_of = [&](const openMVG::Vec3& tstar, const openMVG::Vec3& n) {
        openMVG::Mat3 A;                                 // I - (2 / v) t* n^T
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A(i, j) = (i == j ? 1.0 : 0.0) - (2.0 / v) * tstar[i] * n[j];
        openMVG::Mat3 R = mul(Hn, A);
        if (det3(R) < 0.0) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = -R(i, j);
        return R;
    };
    auto apply = [](const openMVG::Mat3& R, const openMVG::Vec3& a) {
        return openMVG::Vec3(R(0, 0) * a[0] + R(0, 1) * a[1] + R(0, 2) * a[2], R(1, 0) * a[0] + R(1, 1) * a[1] + R(1, 2) * a[2],
                             R(2, 0) * a[0] + R(2, 1) * a[1] + R(2, 2) * a[2]);
    };
    auto neg = [](const openMVG::Vec3& a) { return openMVG::Vec3(-a[0], -a[1], -a[2]); };
    const openMVG::Mat3 Ra = rotation_of(ta_star, na), Rb = rotation_of(tb_star, nb);
    const openMVG::Vec3 ta = apply(Ra, ta_star), tb = apply(Rb, tb_star);
    motions->push_back(HomographyMotion{ Ra, ta, na });
    motions->push_back(HomographyMotion{ Ra, neg(ta), neg(na) });
    motions->push_back(HomographyMotion{ Rb, tb, nb });
    motions->push_back(HomographyMotion{ Rb, neg(tb), neg(nb) });
    return 4;
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

    enum Status { kOk = 0, kEstimateFailed = 1, kModelNotOnGpuPath = 2 /* a model letter RobustMatcher.hpp:399-405 does not know */ };
    Status lastStatus() const { return last_status_; }

    // RobustMatcher::filterFundamental (:128-151): ACRANSAC(ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError, UnnormalizerT>(x1, w, h,
    // x2, w, h, true), inliers, 256, &F, initial_residual_tolerance) -> clc_two_view_acransac(.., 'F', ..); the matrix goes where the
    // reference puts it (relativePose_info.essential_matrix), found_residual_precision is the constant 5.0 of :145, failure iff fewer
    // than 2.5 x 7 inliers; findPose is not used by this model (no pose comes out of F alone in the reference either).
    bool filterFundamental(const openMVG::cameras::IntrinsicBase* intrinsics1, const openMVG::cameras::IntrinsicBase* intrinsics2,
                           const openMVG::Mat& x1, const openMVG::Mat& x2, openMVG::sfm::RelativePose_Info& relativePose_info, colocParams& prm,
                           bool /*findPose*/)
    {
        if (!intrinsics1 || !intrinsics2 || !ctx_) return EXIT_FAILURE;
        double emax = 0.0;
        if (!two_view_filter(CLC_MODEL_FUNDAMENTAL, x1, x2, prm, relativePose_info.initial_residual_tolerance, relativePose_info, &emax))
            return EXIT_FAILURE;
        relativePose_info.found_residual_precision = 5.0;
        if (relativePose_info.vec_inliers.size() < 2.5 * 7) return EXIT_FAILURE;
        return EXIT_SUCCESS;
    }

    // RobustMatcher::decomposeHomography (:106-126): the motions of H under params->K[0], every translation normalised and handed to
    // Pose3 in the place of the CENTRE, as the reference writes it (:123).  Returns EXIT_SUCCESS.
    bool decomposeHomography(openMVG::Mat3& H, std::vector<openMVG::geometry::Pose3>& motions)
    {
        std::vector<hipgeom::HomographyMotion> ms;
        hipgeom::decompose_homography_mat(H, params->K[0], &ms);
        for (const hipgeom::HomographyMotion& m : ms) {
            const double nt = std::sqrt(m.t[0] * m.t[0] + m.t[1] * m.t[1] + m.t[2] * m.t[2]);
            const openMVG::Vec3 tn = nt > 0.0 ? openMVG::Vec3(m.t[0] / nt, m.t[1] / nt, m.t[2] / nt) : m.t;   // (Eigen: normalized() of 0 is 0)
            motions.emplace_back(m.R, tn);
        }
        return EXIT_SUCCESS;
    }

    // RobustMatcher::performChiralityTest (:39-104): the vote of RelativePoseFromEssential over caller-supplied candidates
    bool performChiralityTest(const openMVG::Mat3X& x1, const openMVG::Mat3X& x2, const openMVG::Mat3& /*E*/,
                              const std::vector<uint32_t>& bearing_vector_index_to_use, std::vector<openMVG::geometry::Pose3> relative_poses,
                              openMVG::geometry::Pose3* final_pose, std::vector<uint32_t>* vec_selected_points = nullptr,
                              std::vector<openMVG::Vec3>* vec_points = nullptr, const double positive_depth_solution_ratio = 0.7)
    {
        return hipgeom::chirality_vote(x1, x2, bearing_vector_index_to_use, relative_poses, final_pose, vec_selected_points, vec_points,
                                       positive_depth_solution_ratio);
    }

    // RobustMatcher::filterHomography (:188-239): ACRANSAC(ACKernelAdaptor<FourPointSolver, AsymmetricError, UnnormalizerI>(x1, w, h, x2,
    // w, h, false), inliers, 256, &H, +inf) -> clc_two_view_acransac(.., 'H', ..); failure iff fewer than 2.5 x 4 inliers; otherwise
    // found_residual_precision = the a-contrario threshold and, with findPose, the motions of H voted on by the chirality test -- whose
    // verdict the reference does not look at (:218): relativePose is whatever it left in final_pose.
    bool filterHomography(const openMVG::cameras::IntrinsicBase* intrinsics1, const openMVG::cameras::IntrinsicBase* intrinsics2,
                          const openMVG::Mat& x1, const openMVG::Mat& x2, openMVG::sfm::RelativePose_Info& relativePose_info, colocParams& prm,
                          bool findPose)
    {
        if (!intrinsics1 || !intrinsics2 || !ctx_) return EXIT_FAILURE;
        const openMVG::Mat3X normpt2D_1 = (*intrinsics1)(x1), normpt2D_2 = (*intrinsics2)(x2);
        double emax = 0.0;
        if (!two_view_filter(CLC_MODEL_HOMOGRAPHY, x1, x2, prm, std::numeric_limits<double>::infinity(), relativePose_info, &emax))
            return EXIT_FAILURE;
        if (relativePose_info.vec_inliers.size() < 4 * 2.5) return EXIT_FAILURE;
        relativePose_info.found_residual_precision = emax;
        if (findPose) {
            std::vector<openMVG::geometry::Pose3> motions;
            decomposeHomography(relativePose_info.essential_matrix, motions);
            openMVG::geometry::Pose3 final_pose;
            (void)performChiralityTest(normpt2D_1, normpt2D_2, relativePose_info.essential_matrix, relativePose_info.vec_inliers, motions,
                                       &final_pose);
            relativePose_info.relativePose = final_pose;
        }
        return EXIT_SUCCESS;
    }

    // RobustMatcher::computeRelativePose (:372-424): positions of the pair's putative matches, undistorted through each camera's
    // radial-K3 model, into the filter selected by params->model (:399-405): 'E' (the one the reference's own launch code uses), 'F'
    // or 'H', all three on the GPU; another letter is "Unknown filtering type" (:406-408) -> kModelNotOnGpuPath.
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
        else if (params->model == 'F' || params->model == 'H') {
            status = params->model == 'F' ? filterFundamental(&camL, &camR, xL, xR, relativePose, *params, true)
                                          : filterHomography(&camL, &camR, xL, xR, relativePose, *params, true);
            if (status == EXIT_FAILURE) last_status_ = kEstimateFailed;
        }
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
    // the a-contrario filter of model 'F' / 'H' on the GPU: inliers and the model matrix (pixels) into relativePose_info; false: the call failed
    bool two_view_filter(const int model, const openMVG::Mat& x1, const openMVG::Mat& x2, colocParams& prm, const double precision,
                         openMVG::sfm::RelativePose_Info& relativePose_info, double* emax)
    {
        const int n = static_cast<int>(x1.cols());
        std::vector<double> p1(2 * static_cast<size_t>(n)), p2(2 * static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) {
            p1[2 * i] = x1(0, i); p1[2 * i + 1] = x1(1, i);
            p2[2 * i] = x2(0, i); p2[2 * i + 1] = x2(1, i);
        }
        double M[9] = {}, nfa = 0.0;
        std::vector<int32_t> inl(static_cast<size_t>(n > 0 ? n : 1));
        int n_inl = 0, its = 0;
        const int rc = clc_two_view_acransac(ctx_, model, p1.data(), p2.data(), n, nullptr, nullptr, static_cast<int>(prm.imageSize.first),
                                             static_cast<int>(prm.imageSize.second), iterationCount, seed++, precision, M, nullptr, nullptr,
                                             inl.data(), &n_inl, emax, &nfa, &its);
        if (rc != CLC_OK) {
            std::cerr << "HIPRobustMatcher: clc_two_view_acransac: " << clc_last_error_string(ctx_) << std::endl;
            return false;
        }
        relativePose_info.vec_inliers.assign(inl.begin(), inl.begin() + n_inl);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) relativePose_info.essential_matrix(i, j) = M[3 * i + j];
        return true;
    }
    bool not_on_gpu_path(const char model, const char* where)
    {
        last_status_ = kModelNotOnGpuPath;
        std::cerr << "HIPRobustMatcher: unknown filtering type '" << model << "' (" << where << ")." << std::endl;
        return EXIT_FAILURE;
    }
    Status last_status_ = kOk;
    clc_ctx* ctx_ = nullptr;
    colocParams* params;
};

} // namespace coloc
