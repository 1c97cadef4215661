// inter_pose.hip -- the inter-camera step of ColoC::interPoseEstimator behind the C ABI (include/coloc_hip.h clc_inter_pose_batch;
// reference include/coloc/coloc.hpp:296-340) for several camera pairs at once: the a-contrario five-point filters of all pairs
// (pose_batch.hip), per pair the host geometry (inter_geometry.cpp), then all refinements enqueued on the pairs' contexts and
// collected through pinned records.  Between the two halves of the geometry sits the reference's own way to the features the pair's
// temporary map shares with the global map (coloc.hpp:317-323): the temporary map's descriptors gathered on the device and matched
// against the global map's (K2NN, threshold 60), for the jobs that ask for it.
#include "clc_ctx.h"
#include "inter_geometry.h"

#include <chrono>
#include <cstring>
#include <utility>
#include <vector>

using namespace clc;

namespace {

// one refinement staged and enqueued on the context's stream, its record going to pinned memory; returns the `ready` word to poll
int refine_enqueue(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const double* h_Rt_in, double huber_a,
                   int32_t** ready, double** h_rec)
{
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_d = (size_t)5 * N + 16 + 12;
    const size_t out_d = (pnp_refine_out_bytes() + 7) / 8;
    int rc = ensure_pnp(ctx, in_d + out_d + 8);
    if (rc != CLC_OK) return rc;
    rc = ensure_pinned(ctx, (in_d + out_d) * sizeof(double) + 64);
    if (rc != CLC_OK) return rc;
    double* dX = ctx->d_pnp;
    double* dx = dX + (size_t)3 * N;
    double* dK = dx + (size_t)2 * N;
    double* dRt = dK + 16;
    double* hp = (double*)ctx->h_pin;
    memcpy(hp, h_X, sizeof(double) * 3 * N);
    memcpy(hp + (size_t)3 * N, h_x, sizeof(double) * 2 * N);
    memset(hp + (size_t)5 * N, 0, sizeof(double) * 16);
    memcpy(hp + (size_t)5 * N, h_K, sizeof(double) * 9);
    memcpy(hp + (size_t)5 * N + 16, h_Rt_in, sizeof(double) * 12);
    *h_rec = hp + in_d;
    *ready = (int32_t*)((uint8_t*)*h_rec + pnp_refine_ready_offset());
    __atomic_store_n(*ready, 0, __ATOMIC_RELAXED);
    CLC_HIP(ctx, launch_acr_stage(hp, ctx->d_pnp, (in_d + 1) & ~(size_t)1, ctx->stream));          // inputs by a launch, not a copy command
    CLC_HIP(ctx, launch_pnp_refine(dRt, dX, dx, nullptr, N, dK, huber_a > 0.0 ? huber_a : 16.0, 50, ctx->d_pnp + in_d, ctx->stream, &ctx->prof, nullptr,
                                   *h_rec));
    return CLC_OK;
}

// rows idx[0 .. n) of a descriptor block, 16 bytes per thread
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int32_t* __restrict__ idx, uint4* __restrict__ dst, const uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n * 4u) dst[i] = src[(size_t)idx[i >> 2] * 4u + (i & 3u)];
}

// matchMapFeatures(mapRegions, interMapRegions) for one pair (coloc.hpp:317-323): the temporary map's descriptors -- those of the
// correspondences in front of both cameras, taken from the lower camera's block -- against the global map's; enqueue only, the
// matches land in the context's pinned block: h_match[q] = temporary map point matched by global map point q, or -1
int map_match_enqueue(clc_ctx* ctx, const clc_inter_pose_job& jb, const InterFront& fr, int32_t** h_match)
{
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nf = fr.corr.size(), nm = (size_t)jb.map_n;
    // device: [ idx nf int32 | gathered nf x 64 B | match nm int32 ]   pinned: [ idx | match ]
    const size_t idx_d = (nf * 4 + 15) / 16 * 2, rows_d = nf * 8, match_d = (nm * 4 + 15) / 16 * 2;
    int rc = ensure_pnp(ctx, idx_d + rows_d + match_d + 8);
    if (rc != CLC_OK) return rc;
    rc = ensure_pinned(ctx, (idx_d + match_d) * sizeof(double) + 64);
    if (rc != CLC_OK) return rc;
    int32_t* h_idx = (int32_t*)ctx->h_pin;
    *h_match = (int32_t*)((double*)ctx->h_pin + idx_d);
    for (size_t k = 0; k < nf; ++k) h_idx[k] = jb.first_feature[fr.corr[k]];
    int32_t* d_idx = (int32_t*)ctx->d_pnp;
    uint4* d_rows = (uint4*)(ctx->d_pnp + idx_d);
    int32_t* d_match = (int32_t*)(ctx->d_pnp + idx_d + rows_d);
    CLC_HIP(ctx, hipMemcpyAsync(d_idx, h_idx, nf * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((nf * 4 + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4*)jb.d_first_desc, (const int32_t*)d_idx,
                       d_rows, (uint32_t)nf);
    CLC_HIP(ctx, hipGetLastError());
    std::vector<K2nnJobDev> jobs(1);
    jobs[0] = K2nnJobDev{};
    jobs[0].q = (const uint4*)jb.d_map_desc; jobs[0].t = (const uint4*)d_rows; jobs[0].out = d_match;
    jobs[0].nq = (uint32_t)nm; jobs[0].nt = (uint32_t)nf;
    jobs[0].thr = (uint32_t)(uint8_t)(jb.match_threshold > 0 ? jb.match_threshold : 60);          // GPUMatcher.hpp:162
    rc = run_jobs(ctx, jobs, ctx->stream);
    if (rc != CLC_OK) return rc;
    CLC_HIP(ctx, hipMemcpyAsync(*h_match, d_match, nm * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    return CLC_OK;
}

} // namespace

extern "C" {

int clc_inter_pose_batch(clc_ctx* const* ctxs, clc_inter_pose_job* jobs, int n_jobs)
{
    if (n_jobs < 0 || (n_jobs > 0 && (!ctxs || !jobs))) return CLC_ERR_BAD_ARG;
    if (n_jobs == 0) return CLC_OK;
    const int rc0 = check_batch_contexts(ctxs, n_jobs, "inter_pose_batch: every job needs a context of its own");
    if (rc0 != CLC_OK) return rc0;
    for (int i = 0; i < n_jobs; ++i) {
        clc_inter_pose_job& jb = jobs[i];
        const int chain = (jb.d_first_desc != nullptr) + (jb.first_feature != nullptr) + (jb.d_map_desc != nullptr);
        if (!jb.tv.E || !jb.tv.inliers || !jb.Rt_source || ((jb.map_index || chain) && !jb.map_X) || (chain != 0 && chain != 3) ||
            (chain == 3 && ((((uintptr_t)jb.d_first_desc | (uintptr_t)jb.d_map_desc) & 15u) || jb.map_n <= 0)))
            return fail(ctxs[i], CLC_ERR_BAD_ARG, "inter_pose_batch: a job needs tv.E, tv.inliers, Rt_source, map_X with map_index, and d_first_desc / first_feature / "
                                                  "d_map_desc all three (16-byte aligned) or none");
        memset(jb.Rt, 0, sizeof jb.Rt); memset(jb.cov, 0, sizeof jb.cov);
        jb.rmse = 0.0; jb.scale = 0.0; jb.n_front = 0; jb.n_common = 0; jb.n_refined = 0; jb.n_map_matches = 0; jb.stage = CLC_INTER_NO_MODEL;
    }
    // 1. the a-contrario five-point filters of all pairs, their chains of launches interleaved (or sharing their launches)
    std::vector<clc_two_view_job*> tv((size_t)n_jobs);
    for (int i = 0; i < n_jobs; ++i) tv[(size_t)i] = &jobs[i].tv;
    int worst = acr_two_view_batch(ctxs, tv.data(), n_jobs, 1);
    struct Pending { int job; int32_t* ready; double* rec; };
    std::vector<Pending> pend;
    std::vector<InterFront> fr((size_t)n_jobs);
    std::vector<std::vector<double>> Xw((size_t)n_jobs);
    std::vector<int32_t*> h_match((size_t)n_jobs, nullptr);
    // 2. host geometry, first half: relative pose + temporary map; the reference's chain then puts its map-to-map sweep into the job's stream
    for (int i = 0; i < n_jobs; ++i) {
        clc_inter_pose_job& jb = jobs[i];
        if (jb.tv.status != CLC_OK) continue;
        jb.stage = inter_relative(jb, fr[(size_t)i]);
        if (jb.stage != CLC_INTER_OK || !jb.d_first_desc) continue;
        const int rc = map_match_enqueue(ctxs[i], jb, fr[(size_t)i], &h_match[(size_t)i]);
        if (rc != CLC_OK) { jb.tv.status = rc; jb.stage = CLC_INTER_NO_SCALE; if (worst == CLC_OK) worst = rc; }
    }
    // 3. second half: the common features (from the sweep, or from the caller's map indices), scale, first pose; the refinement enqueued
    for (int i = 0; i < n_jobs; ++i) {
        clc_inter_pose_job& jb = jobs[i];
        if (jb.tv.status != CLC_OK || jb.stage != CLC_INTER_OK) continue;
        const InterFront& f = fr[(size_t)i];
        std::vector<std::pair<int32_t, int32_t>> common;
        if (jb.d_first_desc) {
            const hipError_t e = hipStreamSynchronize(ctxs[i]->stream);
            if (e != hipSuccess) { jb.tv.status = fail(ctxs[i], CLC_ERR_HIP, "inter_pose_batch: map-to-map sweep", e); jb.stage = CLC_INTER_NO_SCALE; if (worst == CLC_OK) worst = jb.tv.status; continue; }
            // commonFeatures = IndMatch(i_ = global map point, j_ = temporary map point), accepted map points in ascending order (GPUMatcher.hpp:217)
            for (int32_t q = 0; q < jb.map_n; ++q) if (h_match[(size_t)i][q] >= 0) common.emplace_back(q, h_match[(size_t)i][q]);
            jb.n_map_matches = (int)common.size();
        } else if (jb.map_index) {
            for (size_t k = 0; k < f.corr.size(); ++k) {
                const int32_t gi = jb.map_index[f.corr[k]];
                if (gi >= 0 && gi < jb.map_n) common.emplace_back(gi, (int32_t)k);
            }
        }
        jb.stage = inter_scale_pose(jb, f, common, Xw[(size_t)i]);
        if (jb.stage != CLC_INTER_OK) continue;
        Pending p{ i, nullptr, nullptr };
        const int rc = refine_enqueue(ctxs[i], Xw[(size_t)i].data(), f.x2f.data(), jb.n_front, jb.tv.K2, jb.Rt, jb.huber_a, &p.ready, &p.rec);
        if (rc != CLC_OK) { jb.tv.status = rc; jb.stage = CLC_INTER_NO_REFINEMENT; if (worst == CLC_OK) worst = rc; continue; }
        pend.push_back(p);
    }
    // 4. collect: poll the pinned records, fall back to the stream synchronisation after 5 ms
    const auto t0 = std::chrono::steady_clock::now();
    for (const Pending& p : pend) {
        clc_inter_pose_job& jb = jobs[p.job];
        uint32_t spins = 0;
        while (__atomic_load_n(p.ready, __ATOMIC_ACQUIRE) == 0) {
            if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
                const hipError_t e = hipStreamSynchronize(ctxs[p.job]->stream);
                if (e != hipSuccess || __atomic_load_n(p.ready, __ATOMIC_ACQUIRE) == 0) {
                    jb.tv.status = fail(ctxs[p.job], CLC_ERR_HIP, "inter_pose_batch: refinement did not complete", e);
                    jb.stage = CLC_INTER_NO_REFINEMENT;
                    if (worst == CLC_OK) worst = jb.tv.status;
                    break;
                }
            }
        }
        if (jb.stage != CLC_INTER_OK) continue;
        struct { double Rt[12]; double cov[36]; double cost; double rmse; int32_t iterations; int32_t n_used; } f;
        memcpy(&f, p.rec, sizeof f);
        memcpy(jb.Rt, f.Rt, sizeof f.Rt);
        memcpy(jb.cov, f.cov, sizeof f.cov);
        jb.rmse = f.rmse;
        jb.n_refined = f.n_used;
    }
    return worst;
}

} // extern "C"
