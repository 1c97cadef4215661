// pnp.hip -- batched-hypothesis reprojection scoring for PnP / RANSAC on gfx950 (MI355X).
//
// What it replaces: inside openMVG::sfm::SfM_Localizer::Localize (called at reference
// include/coloc/Localizer.hpp:93 with P3P_KE_CVPR17, max_iteration = 256, :82-84) every RANSAC
// iteration evaluates the pixel reprojection error of each candidate [R|t] over all N 2D-3D
// correspondences on one CPU thread.  Here ALL H hypotheses (<= 256 iterations x <= 4 P3P roots)
// are scored in one launch: grid = (point tiles) x (hypotheses), the 12 pose doubles and the 9
// intrinsics are wave-uniform scalar loads, points are read coalesced (and stay L2-resident
// across hypotheses: N x 40 B).
//
//   err[h][i] = || x_i - hnormalized( K (R_h X_i + t_h) ) ||^2      (fp64, pixels^2)
//
// fp64 throughout, operation order identical to the oracle (oracle/clc_oracle.c
// orc_pnp_residuals) and no FMA contraction, so the residual matrix is reproduced exactly; the
// fused score kernel reduces in a different order than a sequential sum, hence the 1e-12
// relative tolerance on `cost` in the tests.
#include "clc_internal.h"
#include "p3p.h"
#include "fivept_wave.h"

namespace clc {

__device__ __forceinline__ double reproj_err(const double* __restrict__ P, const double* __restrict__ K,
                                             const double Xw, const double Yw, const double Zw,
                                             const double u_obs, const double v_obs)
{
    const double xc = ((P[0] * Xw + P[1] * Yw) + P[2] * Zw) + P[3];
    const double yc = ((P[4] * Xw + P[5] * Yw) + P[6] * Zw) + P[7];
    const double zc = ((P[8] * Xw + P[9] * Yw) + P[10] * Zw) + P[11];
    const double u = (K[0] * xc + K[1] * yc) + K[2] * zc;
    const double v = (K[3] * xc + K[4] * yc) + K[5] * zc;
    const double w = (K[6] * xc + K[7] * yc) + K[8] * zc;
    const double du = u_obs - u / w;
    const double dv = v_obs - v / w;
    return du * du + dv * dv;
}

__global__ __launch_bounds__(256) void pnp_residual_kernel(const double* __restrict__ Rt, const double* __restrict__ X,
                                                           const double* __restrict__ x, const int N,
                                                           const double* __restrict__ K, double* __restrict__ err)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y;
    if (i >= N) return;
    const double* P = Rt + (size_t)12 * h;
    err[(size_t)h * N + i] = reproj_err(P, K, X[3 * i], X[3 * i + 1], X[3 * i + 2], x[2 * i], x[2 * i + 1]);
}

// one workgroup per hypothesis: inlier count (exact) + truncated cost (tree-reduced)
__global__ __launch_bounds__(256) void pnp_score_kernel(const double* __restrict__ Rt, const double* __restrict__ X,
                                                        const double* __restrict__ x, const int N,
                                                        const double* __restrict__ K, const double thr2,
                                                        int32_t* __restrict__ count, double* __restrict__ cost)
{
    __shared__ double s_cost[256];
    __shared__ int s_cnt[256];
    const int h = blockIdx.x;
    const double* P = Rt + (size_t)12 * h;
    int cnt = 0;
    double c = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const double e = reproj_err(P, K, X[3 * i], X[3 * i + 1], X[3 * i + 2], x[2 * i], x[2 * i + 1]);
        if (e < thr2) { ++cnt; c += e; }
        else c += thr2;
    }
    s_cost[threadIdx.x] = c;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            s_cost[threadIdx.x] += s_cost[threadIdx.x + st];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (count) count[h] = s_cnt[0];
        if (cost) cost[h] = s_cost[0];
    }
}

// ---- batched minimal solver: one P3P problem per FOUR lanes ------------------------------------------
// samples: S x 3 point indices.  The four lanes of a sample all build the quartic and solve it in closed form (the
// same instructions on the same numbers), then lane r polishes root r and turns it into pose slot r of the sample
// (H = 4 S slots).  Slots without a valid pose are NaN so that they score worst (every comparison with NaN is false:
// 0 inliers, cost = N * thr2).  One lane per sample doing the four roots in turn took 13-15 us; the per-root half of
// the chain now runs four wide.
__global__ __launch_bounds__(64) void p3p_kernel(const double* __restrict__ X, const double* __restrict__ x,
                                                 const double* __restrict__ K, const int32_t* __restrict__ samples,
                                                 const int S, const int N, double* __restrict__ Rt,
                                                 const int solve_blocks, const double* __restrict__ stage_src,
                                                 double* __restrict__ stage_dst, const int stage_n,
                                                 const int32_t* __restrict__ n_dev = nullptr)
{
    if ((int)blockIdx.x >= solve_blocks) {
        // staging duty (clc_pnp_ransac / clc_pnp_localize): X, x, K and samples above are the caller's PINNED HOST
        // buffer, read here straight over PCIe; these extra workgroups copy it into device memory for the scoring
        // launches that follow (which read every point a thousand times) -- no separate host-to-device copy command
        const int nb = (int)gridDim.x - solve_blocks;
        for (int i = ((int)blockIdx.x - solve_blocks) * 64 + (int)threadIdx.x; i < stage_n; i += nb * 64) stage_dst[i] = stage_src[i];
        return;
    }
    const int gid = blockIdx.x * 64 + threadIdx.x;
    const int sidx = gid >> 2, root = gid & 3;
    if (sidx >= S || (n_dev && sidx >= *n_dev)) return;
    // slots without a valid pose are NaN (p3p.h p3p_sample_root: the solve of one root, force-inlined here and into acr_round_kernel; the two
    // copies are held to the same bits by tests/test_gpu_acransac.py::test_pose_many_seeds_same_bits_as_p3p_kernel, the gate to re-run
    // on every compiler bump)
    p3p_sample_root(X, x, K, samples[3 * sidx], samples[3 * sidx + 1], samples[3 * sidx + 2], N, root, Rt + (size_t)48 * sidx + 12 * root);
}

// best hypothesis: most inliers, then lowest cost, then lowest index -- and its inlier mask, in ONE
// launch: every workgroup recomputes the (cheap, deterministic) argmax over the <= 64k hypotheses, so no
// inter-workgroup hand-off is needed; workgroup 0 also publishes {h, count, cost, pose}.
struct PnpResult {      // one packed record so the host needs a single D2H copy
    double Rt[12];
    double cost;
    int32_t h;
    int32_t count;
};

__device__ __forceinline__ bool hyp_better(const int32_t* __restrict__ count, const double* __restrict__ cost, int b, int a)
{
    return count[b] > count[a] || (count[b] == count[a] && (cost[b] < cost[a] || (cost[b] == cost[a] && b < a)));
}

__global__ __launch_bounds__(256) void pnp_select_mask_kernel(const double* __restrict__ Rt, const int32_t* __restrict__ count,
                                                              const double* __restrict__ cost, const int H,
                                                              const double* __restrict__ X, const double* __restrict__ x, const int N,
                                                              const double* __restrict__ K, const double thr2,
                                                              uint8_t* __restrict__ mask, PnpResult* __restrict__ res,
                                                              uint8_t* __restrict__ h_mask, PnpResult* __restrict__ h_res)
{
    // h_mask / h_res (nullable): the caller's pinned host buffer -- the record and the mask are written there as well,
    // so that the host needs no device-to-host copy command after the launch
    __shared__ int s_h[256];
    int bh = -1;
    for (int h = threadIdx.x; h < H; h += 256)
        if (bh < 0 || hyp_better(count, cost, h, bh)) bh = h;
    s_h[threadIdx.x] = bh;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            const int a = s_h[threadIdx.x], b = s_h[threadIdx.x + st];
            s_h[threadIdx.x] = a < 0 ? b : ((b >= 0 && hyp_better(count, cost, b, a)) ? b : a);
        }
        __syncthreads();
    }
    const int h = s_h[0];
    const bool ok = h >= 0 && count[h] > 0;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) {
        if (threadIdx.x < 12) {
            const double v = ok ? Rt[(size_t)12 * h + threadIdx.x] : 0.0;
            res->Rt[threadIdx.x] = v;
            if (h_res) h_res->Rt[threadIdx.x] = v;
        }
        if (threadIdx.x == 12) {
            const int32_t hh = ok ? h : -1, cc = ok ? count[h] : 0;
            const double qq = ok ? cost[h] : 0.0;
            res->h = hh; res->count = cc; res->cost = qq;
            if (h_res) { h_res->h = hh; h_res->count = cc; h_res->cost = qq; }
        }
    }
    if (i >= N) return;
    uint8_t m = 0;
    if (ok) {
        const double e = reproj_err(Rt + (size_t)12 * h, K, X[3 * i], X[3 * i + 1], X[3 * i + 2], x[2 * i], x[2 * i + 1]);
        m = e < thr2 ? 1 : 0;
    }
    mask[i] = m;
    if (h_mask) h_mask[i] = m;
}

// ---- single-pose refinement + 6x6 covariance ------------------------
// Role: Localizer::refine -> PoseRefiner::refinePose (reference include/coloc/Localizer.hpp:110-177,
// include/coloc/Refiner.hpp:47-239): minimise 1/2 sum_i rho(||r_i||^2) over the 6 pose parameters
// [angle-axis w | t] (x_cam = R(w) X + t), structure and intrinsics fixed, r_i = observed - projected
// pixel, rho = ceres::HuberLoss(Square(4.0)) (Refiner.hpp:122): rho(s) = s for s <= 256, 2*16*sqrt(s) - 256
// beyond; then the 6x6 covariance block of the pose = (J^T W J)^-1 at the solution (:177-197).
// Ceres itself is absent (OpenMVG third_party, empty submodule) -> unpinned; this is a
// Levenberg-Marquardt on the same cost with the same parametrisation, one workgroup per pose,
// every iteration = one pass over the points (residual + 2x6 Jacobian + 27 sums reduced through
// wave shuffles and LDS) + a 6x6 Cholesky solve by lane 0.
struct RefineOut {
    double Rt[12];
    double cov[36];
    double cost;        // final 1/2 sum rho
    double rmse;        // sqrt(final_cost / (2 n_used))  (Refiner.hpp:226)
    int32_t iterations;
    int32_t n_used;
    int32_t ready;      // written LAST (system-scope release): a host that cleared it in a pinned record can poll it instead of
    int32_t pad_;       // synchronising the stream
};

__device__ __forceinline__ void rodrigues(const double* w, double* R)
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (th2 < 1e-24) {
        R[0] = 1; R[1] = -w[2]; R[2] = w[1]; R[3] = w[2]; R[4] = 1; R[5] = -w[0]; R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
        return;
    }
    const double th = sqrt(th2), ith = 1.0 / th;
    double s, c;
    sincos(th, &s, &c);                 // one shared range reduction
    const double k0 = w[0] * ith, k1 = w[1] * ith, k2 = w[2] * ith, v = 1.0 - c;
    R[0] = c + k0 * k0 * v;      R[1] = k0 * k1 * v - k2 * s; R[2] = k0 * k2 * v + k1 * s;
    R[3] = k1 * k0 * v + k2 * s; R[4] = c + k1 * k1 * v;      R[5] = k1 * k2 * v - k0 * s;
    R[6] = k2 * k0 * v - k1 * s; R[7] = k2 * k1 * v + k0 * s; R[8] = c + k2 * k2 * v;
}

// angle-axis of a rotation matrix (row-major)
__device__ __forceinline__ void log_so3(const double* R, double* w)
{
    const double tr = R[0] + R[4] + R[8];
    double c = 0.5 * (tr - 1.0);
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double th = acos(c);
    const double ax = R[7] - R[5], ay = R[2] - R[6], az = R[3] - R[1];
    if (th < 1e-9) { w[0] = 0.5 * ax; w[1] = 0.5 * ay; w[2] = 0.5 * az; return; }
    if (M_PI - th < 1e-6) {
        // near pi: take the axis from the diagonal
        const double xx = 0.5 * (R[0] + 1.0), yy = 0.5 * (R[4] + 1.0), zz = 0.5 * (R[8] + 1.0);
        double x = sqrt(xx > 0 ? xx : 0), y = sqrt(yy > 0 ? yy : 0), z = sqrt(zz > 0 ? zz : 0);
        if (ax < 0) x = -x; if (ay < 0) y = -y; if (az < 0) z = -z;
        w[0] = th * x; w[1] = th * y; w[2] = th * z;
        return;
    }
    const double f = th / (2.0 * sin(th));
    w[0] = f * ax; w[1] = f * ay; w[2] = f * az;
}

// dR/dw_k (Gallego & Yezzi 2015): (w_k [w]x + [w x (I - R) e_k]x) R / |w|^2 ; generators at w -> 0
__device__ __forceinline__ void d_rodrigues(const double* w, const double* R, double (*dR)[9])
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    for (int k = 0; k < 3; ++k) {
        double A[9];
        if (th2 < 1e-16) {
            for (int i = 0; i < 9; ++i) A[i] = 0.0;
            if (k == 0) { A[5] = -1; A[7] = 1; } else if (k == 1) { A[2] = 1; A[6] = -1; } else { A[1] = -1; A[3] = 1; }
            for (int i = 0; i < 9; ++i) dR[k][i] = A[i];
            continue;
        }
        // u = w x ((I - R) e_k)
        const double m0 = (k == 0 ? 1.0 : 0.0) - R[0 + k], m1 = (k == 1 ? 1.0 : 0.0) - R[3 + k], m2 = (k == 2 ? 1.0 : 0.0) - R[6 + k];
        const double u0 = w[1] * m2 - w[2] * m1, u1 = w[2] * m0 - w[0] * m2, u2 = w[0] * m1 - w[1] * m0;
        const double b0 = w[k] * w[0] + u0, b1 = w[k] * w[1] + u1, b2 = w[k] * w[2] + u2;   // w_k w + u
        // A = [b]x / th2
        A[0] = 0; A[1] = -b2; A[2] = b1; A[3] = b2; A[4] = 0; A[5] = -b0; A[6] = -b1; A[7] = b0; A[8] = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                dR[k][3 * i + j] = (A[3 * i] * R[j] + A[3 * i + 1] * R[3 + j] + A[3 * i + 2] * R[6 + j]) / th2;
    }
}

// One entry of the same derivative: element e (0..8) of dR/dw_k, for 27 lanes working in parallel.
__device__ __forceinline__ double d_rodrigues_entry(const double* w, const double* R, const int k, const int e)
{
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (th2 < 1e-16) {
        const int pos = k == 0 ? 7 : (k == 1 ? 2 : 3), neg = k == 0 ? 5 : (k == 1 ? 6 : 1);
        return e == pos ? 1.0 : (e == neg ? -1.0 : 0.0);
    }
    const double m0 = (k == 0 ? 1.0 : 0.0) - R[0 + k], m1 = (k == 1 ? 1.0 : 0.0) - R[3 + k], m2 = (k == 2 ? 1.0 : 0.0) - R[6 + k];
    const double u0 = w[1] * m2 - w[2] * m1, u1 = w[2] * m0 - w[0] * m2, u2 = w[0] * m1 - w[1] * m0;
    const double b0 = w[k] * w[0] + u0, b1 = w[k] * w[1] + u1, b2 = w[k] * w[2] + u2;
    const double A[9] = { 0, -b2, b1, b2, 0, -b0, -b1, b0, 0 };
    const int i = e / 3, j = e - 3 * i;
    return (A[3 * i] * R[j] + A[3 * i + 1] * R[3 + j] + A[3 * i + 2] * R[6 + j]) / th2;
}

// index of entry (i, j), j >= i, in the packed upper triangle the normal-equation sums are kept in (row-major: 00 01 .. 05 11 ..)
__device__ __forceinline__ constexpr int packed6(const int i, const int j) { return i <= j ? i * 6 - i * (i - 1) / 2 + (j - i) : j * 6 - j * (j - 1) / 2 + (i - j); }

// Cholesky solve of the damped 6x6 system (A + lambda diag(A)) d = g, A given as its PACKED upper triangle (21 values, read
// where they lie -- LDS: a 6 x 6 register copy costs 72 VGPRs in the one lane that runs this); returns false if not SPD
__device__ __forceinline__ bool solve6(const double* Ap, const double* g, double lambda, double* d)
{
    // fully unrolled (compile-time indices) so that L, y stay in registers instead of scratch
    double L[36], Linv[6];    // Linv[i] = 1 / L[i][i]: six divisions instead of twenty-seven
    bool spd = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const double aij = Ap[packed6(i, j)];
            double sum = aij + (i == j ? lambda * (aij > 1e-12 ? aij : 1e-12) : 0.0);
#pragma unroll
            for (int k = 0; k < j; ++k) sum -= L[6 * i + k] * L[6 * j + k];
            if (i == j) { spd = spd && (sum > 0.0); L[6 * i + i] = sqrt(sum > 0.0 ? sum : 1.0); Linv[i] = 1.0 / L[6 * i + i]; }
            else L[6 * i + j] = sum * Linv[j];
        }
    }
    if (!spd) return false;
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double sum = g[i];
#pragma unroll
        for (int k = 0; k < i; ++k) sum -= L[6 * i + k] * y[k];
        y[i] = sum * Linv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double sum = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) sum -= L[6 * k + i] * d[k];
        d[i] = sum * Linv[i];
    }
    return true;
}

// column c of A^-1 (A SPD, packed upper triangle): called by six lanes in parallel, one column each
__device__ __forceinline__ bool invert6_column(const double* A, int c, double* inv)
{
    double e[6], col[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) e[r] = r == c ? 1.0 : 0.0;
    if (!solve6(A, e, 0.0, col)) return false;
#pragma unroll
    for (int r = 0; r < 6; ++r) inv[6 * r + c] = col[r];
    return true;
}

static constexpr int kRefineSums = 29;   // 21 (upper JtWJ) + 6 (JtWr) + 1 (cost) + 1 (points used)
static constexpr int kRefineThreads = 512;

__global__ __launch_bounds__(kRefineThreads) void pnp_refine_kernel(const double* __restrict__ Rt_in, const double* __restrict__ X,
                                                                    const double* __restrict__ x, const uint8_t* __restrict__ mask,
                                                                    const int N, const double* __restrict__ K, const double huber_a,
                                                                    const int max_iter, const int32_t* __restrict__ valid,
                                                                    RefineOut* __restrict__ out_dev, RefineOut* __restrict__ out_host)
{
    // fused multiply-adds in this kernel (the file default is off for the residual kernels compared bit for bit with the oracle;
    // the refinement is checked against scipy / numeric Jacobians to a tolerance)
#pragma clang fp contract(fast)
    RefineOut* const out = out_host ? out_host : out_dev;      // pinned host record (no D2H copy command) or device record
    if (valid && *valid < 0) {          // chained after clc_pnp_ransac that found no pose: nothing to refine
        if (threadIdx.x < 12) out->Rt[threadIdx.x] = 0.0;
        if (threadIdx.x < 36) out->cov[threadIdx.x] = 0.0;
        if (threadIdx.x == 0) { out->cost = 0.0; out->rmse = 0.0; out->iterations = 0; out->n_used = 0; }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&out->ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    extern __shared__ double red[];     // [kRefineSums][kRefineThreads] reduction scratch (dynamic: 116 KB)
    __shared__ double s_par[6], s_try[6], s_R[9], s_dR[3][9], s_part[kRefineSums][16], s_tot[kRefineSums];
    __shared__ double s_A[kRefineSums];   // sums (JtWJ, JtWr, cost, count) at the CURRENT parameters s_par
    __shared__ int s_flag;
    const int tid = threadIdx.x;
    const double fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    const double b2 = huber_a * huber_a;
    if (tid == 0) {
        double R0[9] = { Rt_in[0], Rt_in[1], Rt_in[2], Rt_in[4], Rt_in[5], Rt_in[6], Rt_in[8], Rt_in[9], Rt_in[10] };
        double w[3];
        log_so3(R0, w);
        s_par[0] = w[0]; s_par[1] = w[1]; s_par[2] = w[2]; s_par[3] = Rt_in[3]; s_par[4] = Rt_in[7]; s_par[5] = Rt_in[11];
    }
    __syncthreads();

    // One pass over the points at parameters `par`: cost AND the normal-equation sums, so that an
    // accepted trial step needs no second pass (rejections are rare and only waste the Jacobian part).
    auto pass = [&](const double* par) {
        if (tid < 64) {
            // every lane of the first wave evaluates R (same operands, same result); lane 0 publishes it and lanes
            // 0..26 each one entry of the three derivative matrices -- the serial prologue of a pass is one
            // rotation + one entry instead of one rotation + 27 entries
            double Rl[9];
            rodrigues(par, Rl);
            if (tid < 9) s_R[tid] = Rl[tid];
            if (tid < 27) s_dR[tid / 9][tid % 9] = d_rodrigues_entry(par, Rl, tid / 9, tid % 9);
        }
        __syncthreads();
        double acc[kRefineSums];
#pragma unroll
        for (int i = 0; i < kRefineSums; ++i) acc[i] = 0.0;
        for (int i = tid; i < N; i += kRefineThreads) {
            if (mask && !mask[i]) continue;
            const double X0 = X[3 * i], X1 = X[3 * i + 1], X2 = X[3 * i + 2];
            const double xc = s_R[0] * X0 + s_R[1] * X1 + s_R[2] * X2 + par[3];
            const double yc = s_R[3] * X0 + s_R[4] * X1 + s_R[5] * X2 + par[4];
            const double zc = s_R[6] * X0 + s_R[7] * X1 + s_R[8] * X2 + par[5];
            const double iz = 1.0 / zc, xn = xc * iz, yn = yc * iz;
            const double r0 = x[2 * i] - (fx * xn + sk * yn + cx), r1 = x[2 * i + 1] - (fy * yn + cy);
            const double sq = r0 * r0 + r1 * r1;
            double rho = sq, wgt = 1.0;                                      // rho(s), rho'(s)
            if (sq > b2) {                                                   // Huber tail: rare after RANSAC, and a real branch
                const double r = sqrt(sq);                                   // (skipped when no lane of the wave needs it)
                rho = 2.0 * huber_a * r - b2;
                wgt = huber_a / r;
            }
            acc[27] += 0.5 * rho;
            acc[28] += 1.0;
            // d(proj)/d(xc,yc,zc)
            const double pu0 = fx * iz, pu1 = sk * iz, pu2 = -(fx * xn + sk * yn) * iz;
            const double pv1 = fy * iz, pv2 = -fy * yn * iz;
            double Ju[6], Jv[6];      // Jacobian of the PROJECTION (residual = obs - proj -> J_r = -J)
            // the 27 derivative entries are read from LDS per point (uniform address: a broadcast read); hoisted out of the loop
            // they take 54 VGPRs and push the kernel over its 256-register budget into scratch -- the opaque offset stops that
            int dro = 0;
            asm volatile("" : "+v"(dro));
            const double (*dRp)[9] = reinterpret_cast<const double (*)[9]>(&s_dR[0][0] + dro);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double d0 = dRp[k][0] * X0 + dRp[k][1] * X1 + dRp[k][2] * X2;
                const double d1 = dRp[k][3] * X0 + dRp[k][4] * X1 + dRp[k][5] * X2;
                const double d2 = dRp[k][6] * X0 + dRp[k][7] * X1 + dRp[k][8] * X2;
                Ju[k] = pu0 * d0 + pu1 * d1 + pu2 * d2;
                Jv[k] = pv1 * d1 + pv2 * d2;
            }
            Ju[3] = pu0; Ju[4] = pu1; Ju[5] = pu2;
            Jv[3] = 0.0; Jv[4] = pv1; Jv[5] = pv2;
            int idx = 0;
#pragma unroll
            for (int a_ = 0; a_ < 6; ++a_) {
#pragma unroll
                for (int b_ = a_; b_ < 6; ++b_) acc[idx++] += wgt * (Ju[a_] * Ju[b_] + Jv[a_] * Jv[b_]);
            }
#pragma unroll
            for (int a_ = 0; a_ < 6; ++a_) acc[21 + a_] += wgt * (Ju[a_] * r0 + Jv[a_] * r1);   // = -J_r^T W r : descent rhs
        }
        // Reduction of the 29 sums over the 512 threads THROUGH LDS, transposed: every thread parks its partials in
        // red[sum][thread] (conflict-free 8-byte writes), then 29 x 16 threads each add 32 of them and 29 threads add
        // the 16 partial results.  The shuffle tree it replaces (29 values x 6 levels x 2 ds_bpermute per wave, eight
        // waves on one LDS pipe) took 16-18 k cycles per pass, measured; this takes ~3 k.
#pragma unroll
        for (int i = 0; i < kRefineSums; ++i) red[i * kRefineThreads + tid] = acc[i];
        __syncthreads();
        if (tid < kRefineSums * 16) {
            const int i = tid >> 4, part = tid & 15;
            const double* src = red + i * kRefineThreads + part;
            double v = 0.0;
#pragma unroll 8
            for (int k = 0; k < kRefineThreads / 16; ++k) v += src[16 * k];
            s_part[i][part] = v;
        }
        __syncthreads();
        if (tid < kRefineSums) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) v += s_part[tid][k];
            s_tot[tid] = v;
        }
        __syncthreads();
    };

    // ONE call site of pass() (it is inlined: two copies cost a dozen spilled VGPRs at the 256-register budget of a 512-thread
    // workgroup): the first trip evaluates the start s_par, every later one the trial s_try; the Levenberg-Marquardt bookkeeping
    // around it is the loop  "step -> [flag] -> evaluate -> accept / reject"  unrolled by half a turn.
    double lambda = 1e-4, cost = 0.0;
    int it = 0;
    bool initial = true;
    for (;;) {
        pass(initial ? s_par : s_try);
        if (initial) {
            if (tid < kRefineSums) s_A[tid] = s_tot[tid];
            __syncthreads();
            cost = s_A[27];
            initial = false;
        } else {
            const double new_cost = s_tot[27];
            bool stop = false;
            if (new_cost < cost) {
                const double rel = (cost - new_cost) / fmax(cost, 1e-300);
                if (tid < 6) s_par[tid] = s_try[tid];
                if (tid < kRefineSums) s_A[tid] = s_tot[tid];
                cost = new_cost;
                lambda = fmax(lambda * 0.1, 1e-12);
                __syncthreads();
                if (rel < 1e-8) { ++it; stop = true; }                // function tolerance 1e-8 (Refiner.hpp:169)
            } else {
                lambda *= 10.0;
                __syncthreads();
                if (lambda > 1e10) stop = true;
            }
            if (stop) break;
            ++it;
        }
        // the next step (rejected factorizations only raise lambda and try again)
        bool leave = false;
        for (;;) {
            if (!(it < max_iter)) { leave = true; break; }
            if (tid == 0) {
                double g[6], d[6];
                for (int a_ = 0; a_ < 6; ++a_) g[a_] = s_A[21 + a_];
                bool ok = solve6(s_A, g, lambda, d);
                double gn = 0, dn = 0, pn = 0;
                for (int a_ = 0; a_ < 6; ++a_) { gn = fmax(gn, fabs(g[a_])); dn += d[a_] * d[a_]; pn += s_par[a_] * s_par[a_]; }
                // convergence like the reference's settings (gradient / parameter tolerance 1e-8, Refiner.hpp:169-171)
                s_flag = !ok ? 2 : ((gn < 1e-8 * fmax(1.0, cost) || sqrt(dn) < 1e-8 * (sqrt(pn) + 1e-8)) ? 1 : 0);
                for (int a_ = 0; a_ < 6; ++a_) s_try[a_] = s_par[a_] + (ok ? d[a_] : 0.0);
            }
            __syncthreads();
            const int flag = s_flag;
            if (flag == 1) { leave = true; break; }
            if (flag == 2) {
                lambda *= 10.0;
                if (lambda > 1e10) { leave = true; break; }
                __syncthreads();
                ++it;
                continue;
            }
            break;
        }
        if (leave) break;
    }
    // s_A holds the sums at s_par
    if (tid < 6) {
        if (!invert6_column(s_A, tid, out->cov)) for (int r = 0; r < 6; ++r) out->cov[6 * r + tid] = 0.0;
    }
    if (tid == 0) {
        double R[9];
        rodrigues(s_par, R);
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) out->Rt[4 * i + j] = R[3 * i + j]; out->Rt[4 * i + 3] = s_par[3 + i]; }
        const double n_used = s_A[28];
        out->cost = s_A[27];
        out->rmse = n_used > 0.0 ? sqrt(s_A[27] / (2.0 * n_used)) : 0.0;
        out->iterations = it;
        out->n_used = (int32_t)n_used;
    }
    __threadfence_system();                 // every writer: its stores to the (possibly pinned host) record have left
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&out->ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t launch_pnp_refine(const double* d_Rt_in, const double* d_X, const double* d_x, const uint8_t* d_mask, int N,
                             const double* d_K, double huber_a, int max_iter, void* d_out, hipStream_t stream, Profiler* prof,
                             const int32_t* d_valid, void* h_out)
{
    if (N <= 0) return hipSuccess;
    constexpr size_t kRedBytes = sizeof(double) * kRefineSums * kRefineThreads;
    static bool attr_set[64] = {};      // per device: more than the 64 KB a kernel gets by default
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)pnp_refine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRedBytes);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, true, stream);
    hipLaunchKernelGGL(pnp_refine_kernel, dim3(1), dim3(kRefineThreads), kRedBytes, stream, d_Rt_in, d_X, d_x, d_mask, N, d_K, huber_a, max_iter,
                       d_valid, (RefineOut*)d_out, (RefineOut*)h_out);
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, false, stream);
    return hipGetLastError();
}
size_t pnp_refine_out_bytes() { return sizeof(RefineOut); }
size_t pnp_refine_ready_offset() { return offsetof(RefineOut, ready); }

hipError_t launch_pnp_ransac(const double* d_X, const double* d_x, int N, const double* d_K, const int32_t* d_samples,
                             int S, double thr2, double* d_Rt /* 48*S */, int32_t* d_count, double* d_cost,
                             uint8_t* d_mask, void* d_result, hipStream_t stream, Profiler* prof, const PnpHostStage* hs)
{
    if (S <= 0 || N <= 0) return hipSuccess;
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, true, stream);
    const int solve_blocks = (4 * S + 63) / 64;
    if (hs) {
        // inputs still sit in the pinned host buffer: the solver lanes read their three points from there, extra
        // workgroups stage everything into d_X.. for the launches below
        int stage_blocks = (hs->n_doubles + 255) / 256;
        if (stage_blocks > 64) stage_blocks = 64;
        const double* hX = hs->src;
        const double* hx = hX + (size_t)3 * N;
        const double* hK = hx + (size_t)2 * N;
        const int32_t* hSamples = (const int32_t*)(hK + 16);
        hipLaunchKernelGGL(p3p_kernel, dim3(solve_blocks + stage_blocks), dim3(64), 0, stream, hX, hx, hK, hSamples, S, N, d_Rt,
                           solve_blocks, hs->src, const_cast<double*>(d_X), hs->n_doubles, (const int32_t*)nullptr);
    } else {
        hipLaunchKernelGGL(p3p_kernel, dim3(solve_blocks), dim3(64), 0, stream, d_X, d_x, d_K, d_samples, S, N, d_Rt, solve_blocks,
                           (const double*)nullptr, (double*)nullptr, 0, (const int32_t*)nullptr);
    }
    hipLaunchKernelGGL(pnp_score_kernel, dim3(4 * S), dim3(256), 0, stream, (const double*)d_Rt, d_X, d_x, N, d_K, thr2, d_count, d_cost);
    hipLaunchKernelGGL(pnp_select_mask_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, (const double*)d_Rt,
                       (const int32_t*)d_count, (const double*)d_cost, 4 * S, d_X, d_x, N, d_K, thr2, d_mask, (PnpResult*)d_result,
                       hs ? hs->h_mask : nullptr, hs ? (PnpResult*)hs->h_result : nullptr);
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, false, stream);
    return hipGetLastError();
}

size_t pnp_result_bytes() { return sizeof(PnpResult); }
size_t pnp_result_valid_offset() { return offsetof(PnpResult, h); }

// ---- two-view scoring: symmetric epipolar distance of H fundamental matrices ----------
// The error model RobustMatcher::filterEssential gives AC-RANSAC (reference include/coloc/RobustMatcher.hpp:161-168,
// openMVG SymmetricEpipolarDistanceError on pixel coordinates with F = K2^-T E K1^-1).  Same batched shape as
// the PnP scoring: all hypotheses x all correspondences in one launch; operation order = oracle's (exact).
__device__ __forceinline__ double epipolar_err(const double* __restrict__ f, double u1, double v1, double u2, double v2)
{
    const double a0 = (f[0] * u1 + f[1] * v1) + f[2];
    const double a1 = (f[3] * u1 + f[4] * v1) + f[5];
    const double a2 = (f[6] * u1 + f[7] * v1) + f[8];
    const double b0 = (f[0] * u2 + f[3] * v2) + f[6];
    const double b1 = (f[1] * u2 + f[4] * v2) + f[7];
    const double d = (u2 * a0 + v2 * a1) + a2;
    return (d * d) * (1.0 / (a0 * a0 + a1 * a1) + 1.0 / (b0 * b0 + b1 * b1)) / 4.0;
}

__global__ __launch_bounds__(256) void epipolar_residual_kernel(const double* __restrict__ F, const double* __restrict__ x1,
                                                                const double* __restrict__ x2, const int N, double* __restrict__ err)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    err[(size_t)blockIdx.y * N + i] = epipolar_err(F + (size_t)9 * blockIdx.y, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]);
}

__global__ __launch_bounds__(256) void epipolar_score_kernel(const double* __restrict__ F, const double* __restrict__ x1,
                                                             const double* __restrict__ x2, const int N, const double thr2,
                                                             int32_t* __restrict__ count, double* __restrict__ cost)
{
    __shared__ double s_cost[256];
    __shared__ int s_cnt[256];
    const double* f = F + (size_t)9 * blockIdx.x;
    int cnt = 0;
    double c = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const double e = epipolar_err(f, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]);
        if (e < thr2) { ++cnt; c += e; }
        else c += thr2;
    }
    s_cost[threadIdx.x] = c;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { s_cost[threadIdx.x] += s_cost[threadIdx.x + st]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (count) count[blockIdx.x] = s_cnt[0]; if (cost) cost[blockIdx.x] = s_cost[0]; }
}

// ---- essential-matrix RANSAC: five-point problems, one per lane ---------------------------
// samples: S x 5 correspondence indices.  Each lane normalises its 5 point pairs with K1^-1 / K2^-1, solves the
// five-point problem (csrc/fivept.h, <= 10 real solutions) and writes 10 slots of {F = K2^-T E K1^-1 (9), E (9)};
// unused slots are NaN so that they score worst.  The arrays of the solver live in scratch memory: this kernel is
// latency-bound by design (256 lanes), the scoring that follows is the data-parallel part.
struct EpiResult {      // one packed record so the host needs a single D2H copy
    double E[9];
    double F[9];
    double cost;
    int32_t h;
    int32_t count;
};

__global__ __launch_bounds__(64) void fivept_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                    const double* __restrict__ K1, const double* __restrict__ K2,
                                                    const int32_t* __restrict__ samples, const int S, const int N,
                                                    double* __restrict__ FE /* S x 10 x 18 */, const int32_t* __restrict__ n_dev = nullptr)
{
    // ONE sample per wave: the solver is full of data-dependent steps (pivoting, iterations that stop on convergence), and 64
    // different problems in one wave serialise every divergent branch -- measured 4.8 ms for 256 samples with a problem per
    // lane.  The wave's lanes share the work of their problem instead (csrc/fivept_wave.h); lane 0 writes the result.  The body is
    // fpw::models_of_sample: one not-inlined function shared with the a-contrario round's own solve (acransac.hip).
    const int sidx = blockIdx.x;
    if (sidx >= S || (n_dev && sidx >= *n_dev)) return;
    const int32_t* sp = samples + 5 * sidx;
    fpw::models_of_sample(x1, x2, K1, K2, sp[0], sp[1], sp[2], sp[3], sp[4], N, FE + (size_t)180 * sidx);
}

// score the F part of every slot (stride 18 doubles)
__global__ __launch_bounds__(256) void epipolar_score_strided_kernel(const double* __restrict__ FE, const double* __restrict__ x1,
                                                                     const double* __restrict__ x2, const int N, const double thr2,
                                                                     int32_t* __restrict__ count, double* __restrict__ cost)
{
    __shared__ double s_cost[256];
    __shared__ int s_cnt[256];
    const double* f = FE + (size_t)18 * blockIdx.x;
    int cnt = 0;
    double c = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const double e = epipolar_err(f, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]);
        if (e < thr2) { ++cnt; c += e; }
        else c += thr2;
    }
    s_cost[threadIdx.x] = c;
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { s_cost[threadIdx.x] += s_cost[threadIdx.x + st]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { count[blockIdx.x] = s_cnt[0]; cost[blockIdx.x] = s_cost[0]; }
}

__global__ __launch_bounds__(256) void epipolar_select_mask_kernel(const double* __restrict__ FE, const int32_t* __restrict__ count,
                                                                   const double* __restrict__ cost, const int H,
                                                                   const double* __restrict__ x1, const double* __restrict__ x2,
                                                                   const int N, const double thr2, uint8_t* __restrict__ mask,
                                                                   EpiResult* __restrict__ res)
{
    __shared__ int s_h[256];
    int bh = -1;
    for (int h = threadIdx.x; h < H; h += 256)
        if (bh < 0 || hyp_better(count, cost, h, bh)) bh = h;
    s_h[threadIdx.x] = bh;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            const int a = s_h[threadIdx.x], b = s_h[threadIdx.x + st];
            s_h[threadIdx.x] = a < 0 ? b : ((b >= 0 && hyp_better(count, cost, b, a)) ? b : a);
        }
        __syncthreads();
    }
    const int h = s_h[0];
    const bool ok = h >= 0 && count[h] > 0;
    const double* f = FE + (size_t)18 * (ok ? h : 0);
    if (blockIdx.x == 0) {
        if (threadIdx.x < 9) { res->F[threadIdx.x] = ok ? f[threadIdx.x] : 0.0; res->E[threadIdx.x] = ok ? f[9 + threadIdx.x] : 0.0; }
        if (threadIdx.x == 9) { res->h = ok ? h : -1; res->count = ok ? count[h] : 0; res->cost = ok ? cost[h] : 0.0; }
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    mask[i] = (ok && epipolar_err(f, x1[2 * i], x1[2 * i + 1], x2[2 * i], x2[2 * i + 1]) < thr2) ? 1 : 0;
}

hipError_t launch_essential_ransac(const double* d_x1, const double* d_x2, int N, const double* d_K1, const double* d_K2,
                                   const int32_t* d_samples, int S, double thr2, double* d_FE, int32_t* d_count, double* d_cost,
                                   uint8_t* d_mask, void* d_result, hipStream_t stream, Profiler* prof)
{
    if (S <= 0 || N <= 0) return hipSuccess;
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, true, stream);
    hipLaunchKernelGGL(fivept_kernel, dim3(S), dim3(64), 0, stream, d_x1, d_x2, d_K1, d_K2, d_samples, S, N, d_FE, (const int32_t*)nullptr);
    hipLaunchKernelGGL(epipolar_score_strided_kernel, dim3(10 * S), dim3(256), 0, stream, (const double*)d_FE, d_x1, d_x2, N, thr2,
                       d_count, d_cost);
    hipLaunchKernelGGL(epipolar_select_mask_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, (const double*)d_FE,
                       (const int32_t*)d_count, (const double*)d_cost, 10 * S, d_x1, d_x2, N, thr2, d_mask, (EpiResult*)d_result);
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, false, stream);
    return hipGetLastError();
}
size_t epi_result_bytes() { return sizeof(EpiResult); }

hipError_t launch_epipolar(const double* d_F, int H, const double* d_x1, const double* d_x2, int N, double thr2, double* d_err,
                           int32_t* d_count, double* d_cost, hipStream_t stream, Profiler* prof)
{
    if (H <= 0 || N <= 0) return hipSuccess;
    prof_mark(prof, CLC_KERNEL_PNP_RESIDUALS, true, stream);
    if (d_err) hipLaunchKernelGGL(epipolar_residual_kernel, dim3((N + 255) / 256, H), dim3(256), 0, stream, d_F, d_x1, d_x2, N, d_err);
    else hipLaunchKernelGGL(epipolar_score_kernel, dim3(H), dim3(256), 0, stream, d_F, d_x1, d_x2, N, thr2, d_count, d_cost);
    prof_mark(prof, CLC_KERNEL_PNP_RESIDUALS, false, stream);
    return hipGetLastError();
}

hipError_t launch_pnp_residuals(const double* d_Rt, int H, const double* d_X, const double* d_x, int N,
                                const double* d_K, double* d_err, hipStream_t stream, Profiler* prof)
{
    if (H <= 0 || N <= 0) return hipSuccess;
    prof_mark(prof, CLC_KERNEL_PNP_RESIDUALS, true, stream);
    hipLaunchKernelGGL(pnp_residual_kernel, dim3((N + 255) / 256, H), dim3(256), 0, stream, d_Rt, d_X, d_x, N, d_K, d_err);
    prof_mark(prof, CLC_KERNEL_PNP_RESIDUALS, false, stream);
    return hipGetLastError();
}

hipError_t launch_pnp_score(const double* d_Rt, int H, const double* d_X, const double* d_x, int N,
                            const double* d_K, double thr2, int32_t* d_count, double* d_cost, hipStream_t stream,
                            Profiler* prof)
{
    if (H <= 0) return hipSuccess;
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, true, stream);
    hipLaunchKernelGGL(pnp_score_kernel, dim3(H), dim3(256), 0, stream, d_Rt, d_X, d_x, N, d_K, thr2, d_count, d_cost);
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, false, stream);
    return hipGetLastError();
}

} // namespace clc
