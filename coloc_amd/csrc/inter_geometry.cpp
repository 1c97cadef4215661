// inter_geometry.cpp -- the host part of the inter-camera step (plain C++, no device code) and the fusion arithmetic exported through the
// C ABI.  Reference: ColoC::interPoseEstimator, include/coloc/coloc.hpp:296-340; RobustMatcher.hpp:176-183; colocUtils.hpp:184-211;
// CovIntersection.hpp:24-49.
#include "inter_geometry.h"

#include <algorithm>
#include <cmath>
#include <cstdint>

#include "../host/HIPCovIntersection.hpp"
#include "../host/HIPRobustMatcher.hpp"      // hipgeom::motion_from_essential

namespace clc {

// pixel -> normalised camera plane for an upper-triangular K (row-major)
static inline void normalise_px(const double* K, const double x, const double y, double* n)
{
    n[1] = (y - K[5]) / K[4];
    n[0] = (x - K[2] - K[1] * n[1]) / K[0];
}

// Step 1 of the host part (coloc.hpp:296-306): relative pose from E with the chirality vote (RobustMatcher.hpp:176-183) and the pair's
// temporary map in the source camera's frame (unit baseline) -- the correspondences in front of both cameras.  Triangulation: the
// depths along the two rays that bring them closest (closed form; OpenMVG's TriangulateDLT differs from it by less than the
// measurement noise).  Returns CLC_INTER_OK, or which stage failed.
int inter_relative(clc_inter_pose_job& jb, InterFront& fr)
{
    const clc_two_view_job& tv = jb.tv;
    const int ni = tv.n_inliers;
    jb.n_front = 0; jb.n_common = 0; jb.n_map_matches = 0; jb.scale = 0.0;
    fr.Xt.clear(); fr.x2f.clear(); fr.corr.clear();
    if (ni < 13 || !tv.E || !tv.inliers) return CLC_INTER_NO_MODEL;
    openMVG::Mat3 E;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E(i, j) = tv.E[3 * i + j];
    std::vector<openMVG::geometry::Pose3> cand;
    coloc::hipgeom::motion_from_essential(E, &cand);
    std::vector<double> n1((size_t)2 * ni), n2((size_t)2 * ni);
    for (int k = 0; k < ni; ++k) {
        const int i = tv.inliers[k];
        normalise_px(tv.K1, tv.x1[2 * i], tv.x1[2 * i + 1], &n1[2 * (size_t)k]);
        normalise_px(tv.K2, tv.x2[2 * i], tv.x2[2 * i + 1], &n2[2 * (size_t)k]);
    }
    int best = -1, best_cnt = -1;
    std::vector<double> l1((size_t)ni), bl1;
    std::vector<uint8_t> front((size_t)ni), bfr;
    double Rb[9] = {}, tb[3] = {};
    for (size_t c = 0; c < cand.size(); ++c) {
        const openMVG::Mat3& R = cand[c].rotation();
        const openMVG::Vec3 t = cand[c].translation();
        int cnt = 0;
        for (int k = 0; k < ni; ++k) {
            const double p[3] = { n1[2 * (size_t)k], n1[2 * (size_t)k + 1], 1.0 }, b[3] = { n2[2 * (size_t)k], n2[2 * (size_t)k + 1], 1.0 };
            double a[3];
            for (int r = 0; r < 3; ++r) a[r] = R(r, 0) * p[0] + R(r, 1) * p[1] + R(r, 2) * p[2];
            // min | l1 a - l2 b + t |^2
            const double aa = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], bb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
            const double ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
            const double at = a[0] * t[0] + a[1] * t[1] + a[2] * t[2], bt = b[0] * t[0] + b[1] * t[1] + b[2] * t[2];
            double det = aa * bb - ab * ab;
            if (std::fabs(det) < 1e-18) det = 1e-18;
            const double d1 = (-at * bb + bt * ab) / det, d2 = (-at * ab + bt * aa) / det;
            l1[(size_t)k] = d1;
            front[(size_t)k] = d1 > 0.0 && d2 > 0.0;
            cnt += front[(size_t)k];
        }
        if (cnt > best_cnt) {
            best_cnt = cnt; best = (int)c; bl1 = l1; bfr = front;
            for (int r = 0; r < 3; ++r) { for (int q = 0; q < 3; ++q) Rb[3 * r + q] = R(r, q); tb[r] = t[r]; }
        }
    }
    if (best < 0 || best_cnt < 8) return CLC_INTER_NO_RELATIVE_POSE;
    jb.n_front = best_cnt;
    // the temporary map (source camera's frame, unit baseline) of the correspondences in front of both cameras
    const size_t nf = (size_t)best_cnt;
    fr.Xt.resize(3 * nf); fr.x2f.resize(2 * nf); fr.corr.resize(nf);
    size_t w = 0;
    for (int k = 0; k < ni; ++k) {
        if (!bfr[(size_t)k]) continue;
        const int i = tv.inliers[k];
        fr.Xt[3 * w] = n1[2 * (size_t)k] * bl1[(size_t)k]; fr.Xt[3 * w + 1] = n1[2 * (size_t)k + 1] * bl1[(size_t)k]; fr.Xt[3 * w + 2] = bl1[(size_t)k];
        fr.x2f[2 * w] = tv.x2[2 * i]; fr.x2f[2 * w + 1] = tv.x2[2 * i + 1];
        fr.corr[w] = i;
        ++w;
    }
    for (int r = 0; r < 9; ++r) fr.R[r] = Rb[r];
    for (int r = 0; r < 3; ++r) fr.t[r] = tb[r];
    return CLC_INTER_OK;
}

// Step 2 (coloc.hpp:323-340): the temporary map's scale against the global map through the features both hold -- `common` = pairs (global
// map point, temporary map point) IN THE ORDER the rule walks them -- by the reference's rule (colocUtils.hpp:184-211: the mean over
// consecutive common features of |X12 - X11| / |X22 - X21|, the two norms rounded to float as there), behind a depth-ratio screen that is
// ours (a wrong descriptor match would otherwise enter the mean twice; the reference has no such guard), then the destination's first
// pose through the source's and the temporary map in world coordinates.
int inter_scale_pose(clc_inter_pose_job& jb, const InterFront& fr, const std::vector<std::pair<int32_t, int32_t>>& common, std::vector<double>& Xw)
{
    const std::vector<double>& Xt = fr.Xt;
    const size_t nf = fr.corr.size();
    const double* Rb = fr.R; const double* tb = fr.t;
    const double* Rs = jb.Rt_source;          // [R|t] row-major 3 x 4
    std::vector<size_t> com;                  // positions in `common`
    std::vector<double> ratio;
    for (size_t c = 0; c < common.size(); ++c) {
        const int32_t gi = common[c].first;
        const size_t k = (size_t)common[c].second;
        if (gi < 0 || gi >= jb.map_n || k >= nf) continue;
        const double* Xg = jb.map_X + 3 * (size_t)gi;
        double xs[3];
        for (int r = 0; r < 3; ++r) xs[r] = Rs[4 * r] * Xg[0] + Rs[4 * r + 1] * Xg[1] + Rs[4 * r + 2] * Xg[2] + Rs[4 * r + 3];
        const double ng = std::sqrt(xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2]);
        const double nt = std::sqrt(Xt[3 * k] * Xt[3 * k] + Xt[3 * k + 1] * Xt[3 * k + 1] + Xt[3 * k + 2] * Xt[3 * k + 2]);
        com.push_back(c);
        ratio.push_back(ng / (nt > 1e-12 ? nt : 1e-12));
    }
    jb.n_common = (int)com.size();
    if (com.size() < 8) return CLC_INTER_NO_SCALE;
    std::vector<double> srt(ratio);
    std::sort(srt.begin(), srt.end());
    const double med = (srt.size() & 1) ? srt[srt.size() / 2] : 0.5 * (srt[srt.size() / 2 - 1] + srt[srt.size() / 2]);
    std::vector<size_t> keep;
    for (size_t k = 0; k < com.size(); ++k) if (std::fabs(ratio[k] / med - 1.0) < 0.2) keep.push_back(com[k]);
    jb.n_common = (int)keep.size();
    if (keep.size() < 8) return CLC_INTER_NO_SCALE;
    double sum = 0.0; size_t good = 0;
    for (size_t k = 0; k + 1 < keep.size(); ++k) {
        const double* g0 = jb.map_X + 3 * (size_t)common[keep[k]].first; const double* g1 = jb.map_X + 3 * (size_t)common[keep[k + 1]].first;
        const double* t0 = &Xt[3 * (size_t)common[keep[k]].second]; const double* t1 = &Xt[3 * (size_t)common[keep[k + 1]].second];
        // colocUtils.hpp:201-204: float dist1 = (X12 - X11).norm(); float dist2 = (X22 - X21).norm(); scale += dist1 / dist2;
        const float d1 = (float)std::sqrt((g1[0] - g0[0]) * (g1[0] - g0[0]) + (g1[1] - g0[1]) * (g1[1] - g0[1]) + (g1[2] - g0[2]) * (g1[2] - g0[2]));
        const float d2 = (float)std::sqrt((t1[0] - t0[0]) * (t1[0] - t0[0]) + (t1[1] - t0[1]) * (t1[1] - t0[1]) + (t1[2] - t0[2]) * (t1[2] - t0[2]));
        if (d2 > 1e-9f) { sum += (double)(d1 / d2); ++good; }
    }
    if (good == 0) return CLC_INTER_NO_SCALE;
    const double scale = sum / (double)good;
    if (!(scale > 0.0) || !std::isfinite(scale)) return CLC_INTER_NO_SCALE;
    jb.scale = scale;
    // the destination's pose through the source: X_d = R_rel X_s + s t_rel, X_s = R_s X_w + t_s
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 3; ++q) jb.Rt[4 * r + q] = Rb[3 * r] * Rs[q] + Rb[3 * r + 1] * Rs[4 + q] + Rb[3 * r + 2] * Rs[8 + q];
        jb.Rt[4 * r + 3] = Rb[3 * r] * Rs[3] + Rb[3 * r + 1] * Rs[7] + Rb[3 * r + 2] * Rs[11] + scale * tb[r];
    }
    // the temporary map in world coordinates: X_w = R_s^T (s X_tmp - t_s)
    Xw.resize(3 * nf);
    for (size_t k = 0; k < nf; ++k) {
        const double v[3] = { scale * Xt[3 * k] - Rs[3], scale * Xt[3 * k + 1] - Rs[7], scale * Xt[3 * k + 2] - Rs[11] };
        for (int q = 0; q < 3; ++q) Xw[3 * k + q] = Rs[q] * v[0] + Rs[4 + q] * v[1] + Rs[8 + q] * v[2];
    }
    return CLC_INTER_OK;
}

} // namespace clc

extern "C" {

int clc_cov_intersection(const double* CA, const double* CB, const double* ca, const double* cb, double* omega,
                         double* cov_fused, double* pos_fused)
{
    if (!CA || !CB || !ca || !cb) return CLC_ERR_BAD_ARG;
    coloc::Mat3d A, B;
    coloc::Vec3d a, b;
    for (int i = 0; i < 9; ++i) { A[i] = CA[i]; B[i] = CB[i]; }
    for (int i = 0; i < 3; ++i) { a[i] = ca[i]; b[i] = cb[i]; }
    coloc::HIPCovIntersection ci;
    ci.loadData(A, B, a, b);
    ci.optimize();
    ci.computeFusedValues();
    if (omega) *omega = ci.minX;
    if (cov_fused) for (int i = 0; i < 9; ++i) cov_fused[i] = ci.covFused[i];
    if (pos_fused) for (int i = 0; i < 3; ++i) pos_fused[i] = ci.poseFused[i];
    return CLC_OK;
}

} // extern "C"
