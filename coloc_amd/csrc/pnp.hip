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

// ---- batched minimal solver: one P3P problem per lane -------------------------------------------
// samples: S x 3 point indices.  Writes 4 pose slots per sample (H = 4 S); unused slots are NaN so
// that they score worst (every comparison with NaN is false: 0 inliers, cost = N * thr2).
__global__ __launch_bounds__(64) void p3p_kernel(const double* __restrict__ X, const double* __restrict__ x,
                                                 const double* __restrict__ K, const int32_t* __restrict__ samples,
                                                 const int S, const int N, double* __restrict__ Rt)
{
    const int sidx = blockIdx.x * 64 + threadIdx.x;
    if (sidx >= S) return;
    double Xs[3][3], f[3][3];
    bool ok = true;
    const double fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int i = samples[3 * sidx + p];
        if (i < 0 || i >= N) { ok = false; i = 0; }
        Xs[p][0] = X[3 * i]; Xs[p][1] = X[3 * i + 1]; Xs[p][2] = X[3 * i + 2];
        const double yn = (x[2 * i + 1] - cy) / fy;
        const double xn = (x[2 * i] - cx - sk * yn) / fx;
        const double nrm = sqrt(xn * xn + yn * yn + 1.0);
        f[p][0] = xn / nrm; f[p][1] = yn / nrm; f[p][2] = 1.0 / nrm;
    }
    double sol[48];
    const int n = ok ? p3p_solve(Xs, f, sol) : 0;
    double* out = Rt + (size_t)48 * sidx;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 12; ++e) out[12 * k + e] = k < n ? sol[12 * k + e] : qnan;
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
                                                              uint8_t* __restrict__ mask, PnpResult* __restrict__ res)
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
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) {
        if (threadIdx.x < 12) res->Rt[threadIdx.x] = ok ? Rt[(size_t)12 * h + threadIdx.x] : 0.0;
        if (threadIdx.x == 12) { res->h = ok ? h : -1; res->count = ok ? count[h] : 0; res->cost = ok ? cost[h] : 0.0; }
    }
    if (i >= N) return;
    if (!ok) { mask[i] = 0; return; }
    const double e = reproj_err(Rt + (size_t)12 * h, K, X[3 * i], X[3 * i + 1], X[3 * i + 2], x[2 * i], x[2 * i + 1]);
    mask[i] = e < thr2 ? 1 : 0;
}

hipError_t launch_pnp_ransac(const double* d_X, const double* d_x, int N, const double* d_K, const int32_t* d_samples,
                             int S, double thr2, double* d_Rt /* 48*S */, int32_t* d_count, double* d_cost,
                             uint8_t* d_mask, void* d_result, hipStream_t stream, Profiler* prof)
{
    if (S <= 0 || N <= 0) return hipSuccess;
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, true, stream);
    hipLaunchKernelGGL(p3p_kernel, dim3((S + 63) / 64), dim3(64), 0, stream, d_X, d_x, d_K, d_samples, S, N, d_Rt);
    hipLaunchKernelGGL(pnp_score_kernel, dim3(4 * S), dim3(256), 0, stream, (const double*)d_Rt, d_X, d_x, N, d_K, thr2, d_count, d_cost);
    hipLaunchKernelGGL(pnp_select_mask_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, (const double*)d_Rt,
                       (const int32_t*)d_count, (const double*)d_cost, 4 * S, d_X, d_x, N, d_K, thr2, d_mask, (PnpResult*)d_result);
    prof_mark(prof, CLC_KERNEL_PNP_SCORE, false, stream);
    return hipGetLastError();
}

size_t pnp_result_bytes() { return sizeof(PnpResult); }

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
