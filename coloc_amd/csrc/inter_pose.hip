// inter_pose.hip -- the inter-camera step of ColoC::interPoseEstimator behind the C ABI (include/coloc_hip.h clc_inter_pose_batch;
// reference include/coloc/coloc.hpp:296-340) for several camera pairs at once: the a-contrario five-point filters of all pairs
// (pose_batch.hip), per pair the host geometry (inter_geometry.cpp), then all refinements enqueued on the pairs' contexts and
// collected through pinned records.
#include "clc_ctx.h"
#include "inter_geometry.h"

#include <chrono>
#include <cstring>
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
        if (!jb.tv.E || !jb.tv.inliers || !jb.Rt_source || (jb.map_index && !jb.map_X))
            return fail(ctxs[i], CLC_ERR_BAD_ARG, "inter_pose_batch: a job needs tv.E, tv.inliers, Rt_source (and map_X with map_index)");
        memset(jb.Rt, 0, sizeof jb.Rt); memset(jb.cov, 0, sizeof jb.cov);
        jb.rmse = 0.0; jb.scale = 0.0; jb.n_front = 0; jb.n_common = 0; jb.n_refined = 0; jb.stage = CLC_INTER_NO_MODEL;
    }
    // 1. the a-contrario five-point filters of all pairs, their chains of launches interleaved (or sharing their launches)
    std::vector<clc_two_view_job*> tv((size_t)n_jobs);
    for (int i = 0; i < n_jobs; ++i) tv[(size_t)i] = &jobs[i].tv;
    int worst = acr_two_view_batch(ctxs, tv.data(), n_jobs);
    struct Pending { int job; int32_t* ready; double* rec; };
    std::vector<Pending> pend;
    std::vector<std::vector<double>> Xw((size_t)n_jobs), x2f((size_t)n_jobs);
    for (int i = 0; i < n_jobs; ++i) {
        clc_inter_pose_job& jb = jobs[i];
        if (jb.tv.status != CLC_OK) continue;
        // 2. host geometry, 3. the refinement enqueued on the job's own context
        jb.stage = inter_geometry(jb, Xw[(size_t)i], x2f[(size_t)i]);
        if (jb.stage != CLC_INTER_OK) continue;
        Pending p{ i, nullptr, nullptr };
        const int rc = refine_enqueue(ctxs[i], Xw[(size_t)i].data(), x2f[(size_t)i].data(), jb.n_front, jb.tv.K2, jb.Rt, jb.huber_a, &p.ready, &p.rec);
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
