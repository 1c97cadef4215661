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
