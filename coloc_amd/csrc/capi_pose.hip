// capi_pose.hip -- pose scoring and the fixed-threshold robust solvers behind the C ABI (include/coloc_hip.h): residuals / scores of
// caller-supplied hypotheses, P3P and five-point RANSAC with a given threshold, the single-pose refinement.  The data-parallel core of
// SfM_Localizer::Localize (reference include/coloc/Localizer.hpp:82-93), RobustMatcher::filterEssential (RobustMatcher.hpp:153-186) and
// PoseRefiner::refinePose (Refiner.hpp:47-239); the a-contrario rule the reference actually runs is in pose_batch.hip.
#include "clc_ctx.h"

#include <cmath>
#include <cstring>
#include <vector>

using namespace clc;

extern "C" {

/* ---- PnP ------------------------------------------------------------------------------------ */

static int pnp_upload(clc_ctx* ctx, const double* h_Rt, int H, const double* h_X, const double* h_x, int N,
                      const double* h_K, size_t extra, double** dRt, double** dX, double** dx, double** dK, double** dExtra)
{
    const size_t need = (size_t)12 * H + (size_t)5 * N + 16 + extra;
    const int rc = ensure_pnp(ctx, need);
    if (rc != CLC_OK) return rc;
    double* p = ctx->d_pnp;
    *dRt = p; p += (size_t)12 * H;
    *dX = p; p += (size_t)3 * N;
    *dx = p; p += (size_t)2 * N;
    *dK = p; p += 16;
    *dExtra = p;
    CLC_HIP(ctx, hipMemcpyAsync(*dRt, h_Rt, sizeof(double) * 12 * H, hipMemcpyHostToDevice, ctx->stream));
    if (N > 0) {
        CLC_HIP(ctx, hipMemcpyAsync(*dX, h_X, sizeof(double) * 3 * N, hipMemcpyHostToDevice, ctx->stream));
        CLC_HIP(ctx, hipMemcpyAsync(*dx, h_x, sizeof(double) * 2 * N, hipMemcpyHostToDevice, ctx->stream));
    }
    CLC_HIP(ctx, hipMemcpyAsync(*dK, h_K, sizeof(double) * 9, hipMemcpyHostToDevice, ctx->stream));
    return CLC_OK;
}

int clc_pnp_residuals(clc_ctx* ctx, const double* h_Rt, int H, const double* h_X, const double* h_x, int N,
                      const double* h_K, double* h_err)
{
    if (!ctx || H < 0 || N < 0 || !h_K || (H > 0 && !h_Rt) || (N > 0 && (!h_X || !h_x)) || (H > 0 && N > 0 && !h_err))
        return fail(ctx, CLC_ERR_BAD_ARG, "pnp_residuals: bad argument");
    if (H == 0 || N == 0) return CLC_OK;
    if (H > 65535) return fail(ctx, CLC_ERR_CAPACITY, "pnp_residuals: more than 65535 hypotheses per call");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    double *dRt, *dX, *dx, *dK, *dE;
    const int rc = pnp_upload(ctx, h_Rt, H, h_X, h_x, N, h_K, (size_t)H * N, &dRt, &dX, &dx, &dK, &dE);
    if (rc != CLC_OK) return rc;
    CLC_HIP(ctx, launch_pnp_residuals(dRt, H, dX, dx, N, dK, dE, ctx->stream, &ctx->prof));
    CLC_HIP(ctx, hipMemcpyAsync(h_err, dE, sizeof(double) * (size_t)H * N, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

int clc_pnp_score(clc_ctx* ctx, const double* h_Rt, int H, const double* h_X, const double* h_x, int N,
                  const double* h_K, double thr2, int32_t* h_count, double* h_cost)
{
    if (!ctx || H < 0 || N < 0 || !h_K || (H > 0 && !h_Rt) || (N > 0 && (!h_X || !h_x)))
        return fail(ctx, CLC_ERR_BAD_ARG, "pnp_score: bad argument");
    if (H == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    double *dRt, *dX, *dx, *dK, *dE;
    const int rc = pnp_upload(ctx, h_Rt, H, h_X, h_x, N, h_K, (size_t)2 * H, &dRt, &dX, &dx, &dK, &dE);
    if (rc != CLC_OK) return rc;
    double* d_cost = dE;
    int32_t* d_count = (int32_t*)(dE + H);
    CLC_HIP(ctx, launch_pnp_score(dRt, H, dX, dx, N, dK, thr2, d_count, d_cost, ctx->stream, &ctx->prof));
    if (h_cost) CLC_HIP(ctx, hipMemcpyAsync(h_cost, d_cost, sizeof(double) * H, hipMemcpyDeviceToHost, ctx->stream));
    if (h_count) CLC_HIP(ctx, hipMemcpyAsync(h_count, d_count, sizeof(int32_t) * H, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

static int epipolar_impl(clc_ctx* ctx, const double* h_F, int H, const double* h_x1, const double* h_x2, int N, double thr2,
                         double* h_err, int32_t* h_count, double* h_cost)
{
    if (!ctx || H < 0 || N < 0 || (H > 0 && !h_F) || (N > 0 && (!h_x1 || !h_x2))) return fail(ctx, CLC_ERR_BAD_ARG, "epipolar: bad argument");
    if (H == 0 || (N == 0 && h_err)) return CLC_OK;
    if (H > 65535) return fail(ctx, CLC_ERR_CAPACITY, "epipolar: more than 65535 hypotheses per call");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t out = h_err ? (size_t)H * N : (size_t)2 * H;
    const int rc = ensure_pnp(ctx, (size_t)9 * H + (size_t)4 * N + out + 8);
    if (rc != CLC_OK) return rc;
    double* dF = ctx->d_pnp;
    double* d1 = dF + (size_t)9 * H;
    double* d2 = d1 + (size_t)2 * N;
    double* dO = d2 + (size_t)2 * N;
    CLC_HIP(ctx, hipMemcpyAsync(dF, h_F, sizeof(double) * 9 * H, hipMemcpyHostToDevice, ctx->stream));
    if (N > 0) {
        CLC_HIP(ctx, hipMemcpyAsync(d1, h_x1, sizeof(double) * 2 * N, hipMemcpyHostToDevice, ctx->stream));
        CLC_HIP(ctx, hipMemcpyAsync(d2, h_x2, sizeof(double) * 2 * N, hipMemcpyHostToDevice, ctx->stream));
    }
    if (h_err) {
        CLC_HIP(ctx, launch_epipolar(dF, H, d1, d2, N, thr2, dO, nullptr, nullptr, ctx->stream, &ctx->prof));
        CLC_HIP(ctx, hipMemcpyAsync(h_err, dO, sizeof(double) * (size_t)H * N, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        int32_t* dC = (int32_t*)(dO + H);
        CLC_HIP(ctx, launch_epipolar(dF, H, d1, d2, N, thr2, nullptr, dC, dO, ctx->stream, &ctx->prof));
        if (h_cost) CLC_HIP(ctx, hipMemcpyAsync(h_cost, dO, sizeof(double) * H, hipMemcpyDeviceToHost, ctx->stream));
        if (h_count) CLC_HIP(ctx, hipMemcpyAsync(h_count, dC, sizeof(int32_t) * H, hipMemcpyDeviceToHost, ctx->stream));
    }
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

// S minimal samples of `k` distinct indices in [0, N) from a xorshift64* stream
static void draw_samples(uint64_t seed, int S, int N, std::vector<int32_t>& out, int k = 3)
{
    out.resize((size_t)k * S);
    uint64_t st = seed ? seed : 0x9E3779B97F4A7C15ull;
    auto next = [&]() { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st * 0x2545F4914F6CDD1Dull; };   // xorshift64*
    for (int s = 0; s < S; ++s) {
        for (int j = 0; j < k; ++j) {
            int32_t v;
            bool again;
            do {
                v = (int32_t)(next() % (uint64_t)N);
                again = false;
                for (int m = 0; m < j && N > j; ++m) again = again || out[(size_t)k * s + m] == v;
            } while (again);
            out[(size_t)k * s + j] = v;
        }
    }
}

static int pnp_ransac_impl(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K,
                           const int32_t* h_samples, int S, uint64_t seed, double thr2, double* h_Rt, uint8_t* h_mask,
                           int* n_inliers, double* cost, double* h_all_Rt, double refine_huber = -1.0, double* h_cov = nullptr,
                           double* rmse = nullptr)
{
    if (!ctx || N < 0 || S < 0 || !h_K || (N > 0 && (!h_X || !h_x))) return fail(ctx, CLC_ERR_BAD_ARG, "pnp_ransac: bad argument");
    if (n_inliers) *n_inliers = 0;
    if (N < 3 || S == 0) { if (h_mask && N > 0) memset(h_mask, 0, (size_t)N); return CLC_OK; }
    if (S > 16384) return fail(ctx, CLC_ERR_CAPACITY, "pnp_ransac: more than 16384 samples per call");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> drawn;
    if (!h_samples) { draw_samples(seed, S, N, drawn); h_samples = drawn.data(); }
    // device workspace, in doubles:  [ X 3N | x 2N | K 16 | samples (3S int32) ]  <- one H2D copy
    //                               [ Rt 48S | cost 4S | count (4S int32) ]      scratch
    //                               [ result record | mask N bytes ]             <- one D2H copy
    const size_t in_d = (size_t)5 * N + 16 + ((size_t)3 * S + 1) / 2;
    const size_t scr_d = (size_t)48 * S + (size_t)4 * S + ((size_t)4 * S + 1) / 2;
    const bool refine = refine_huber > 0.0;
    const size_t ref_d = refine ? (pnp_refine_out_bytes() + 7) / 8 : 0;      // refine record rides in front of the mask
    const size_t res_d = (pnp_result_bytes() + 7) / 8 + ref_d;
    const size_t out_d = res_d + ((size_t)N + 7) / 8;
    int rc = ensure_pnp(ctx, in_d + scr_d + out_d + 8);
    if (rc != CLC_OK) return rc;
    rc = ensure_pinned(ctx, (in_d + out_d) * sizeof(double) + 64);
    if (rc != CLC_OK) return rc;
    double* dX = ctx->d_pnp;
    double* dx = dX + (size_t)3 * N;
    double* dK = dx + (size_t)2 * N;
    int32_t* dSamples = (int32_t*)(dK + 16);
    double* dRt = ctx->d_pnp + in_d;
    double* dCost = dRt + (size_t)48 * S;
    int32_t* dCount = (int32_t*)(dCost + (size_t)4 * S);
    double* dRes = ctx->d_pnp + in_d + scr_d;
    uint8_t* dMask = (uint8_t*)(dRes + res_d);
    // Inputs go into the pinned buffer and stay there: the first launch reads them over PCIe and stages them into
    // device memory itself, the last launch writes record + mask back into the pinned buffer.  No copy commands: a
    // pose solve is three (four with refinement) kernel launches and one stream synchronisation.
    double* hp = (double*)ctx->h_pin;
    double* hout = hp + in_d;
    memcpy(hp, h_X, sizeof(double) * 3 * N);
    memcpy(hp + (size_t)3 * N, h_x, sizeof(double) * 2 * N);
    memcpy(hp + (size_t)5 * N, h_K, sizeof(double) * 9);
    memcpy(hp + (size_t)5 * N + 16, h_samples, sizeof(int32_t) * 3 * S);
    const size_t ref_off = (pnp_result_bytes() + 7) / 8;
    PnpHostStage hs;
    hs.src = hp;
    hs.n_doubles = (int)in_d;
    hs.h_mask = (uint8_t*)(hout + res_d);
    hs.h_result = hout;
    CLC_HIP(ctx, launch_pnp_ransac(dX, dx, N, dK, dSamples, S, thr2, dRt, dCount, dCost, dMask, dRes, ctx->stream, &ctx->prof, &hs));
    if (refine)
        CLC_HIP(ctx, launch_pnp_refine(dRes /* PnpResult.Rt */, dX, dx, dMask, N, dK, refine_huber, 50, dRes + ref_off, ctx->stream,
                                       &ctx->prof, (const int32_t*)((const uint8_t*)dRes + pnp_result_valid_offset()), hout + ref_off));
    if (h_all_Rt) CLC_HIP(ctx, hipMemcpyAsync(h_all_Rt, dRt, sizeof(double) * 48 * S, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    struct { double Rt[12]; double cost; int32_t h; int32_t count; } r;
    memcpy(&r, hout, sizeof r);
    if (h_Rt) memcpy(h_Rt, r.Rt, sizeof(double) * 12);
    if (h_mask) memcpy(h_mask, (const uint8_t*)(hout + res_d), (size_t)N);
    if (n_inliers) *n_inliers = r.h >= 0 ? r.count : 0;
    if (cost) *cost = r.cost;
    if (refine) {
        struct { double Rt[12]; double cov[36]; double cost; double rmse; int32_t iterations; int32_t n_used; } f;
        memcpy(&f, hout + ref_off, sizeof f);
        if (h_Rt) memcpy(h_Rt, f.Rt, sizeof f.Rt);
        if (h_cov) memcpy(h_cov, f.cov, sizeof f.cov);
        if (rmse) *rmse = f.rmse;
    }
    return CLC_OK;
}

int clc_epipolar_residuals(clc_ctx* ctx, const double* h_F, int H, const double* h_x1, const double* h_x2, int N, double* h_err)
{
    if (H > 0 && N > 0 && !h_err) return fail(ctx, CLC_ERR_BAD_ARG, "epipolar_residuals: null output");
    return epipolar_impl(ctx, h_F, H, h_x1, h_x2, N, 0.0, h_err, nullptr, nullptr);
}

int clc_epipolar_score(clc_ctx* ctx, const double* h_F, int H, const double* h_x1, const double* h_x2, int N, double thr2,
                       int32_t* h_count, double* h_cost)
{
    return epipolar_impl(ctx, h_F, H, h_x1, h_x2, N, thr2, nullptr, h_count, h_cost);
}

static int essential_impl(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1, const double* h_K2,
                          const int32_t* h_samples, int S, uint64_t seed, double thr2, double* h_E, double* h_F, uint8_t* h_mask,
                          int* n_inliers, double* h_all_E)
{
    if (!ctx || N < 0 || S < 0 || !h_K1 || !h_K2 || (N > 0 && (!h_x1 || !h_x2))) return fail(ctx, CLC_ERR_BAD_ARG, "essential_ransac: bad argument");
    if (n_inliers) *n_inliers = 0;
    if (N < 5 || S == 0) { if (h_mask && N > 0) memset(h_mask, 0, (size_t)N); return CLC_OK; }
    if (S > 6000) return fail(ctx, CLC_ERR_CAPACITY, "essential_ransac: more than 6000 samples per call");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> drawn;
    if (!h_samples) { draw_samples(seed, S, N, drawn, 5); h_samples = drawn.data(); }
    // device workspace in doubles: [ x1 2N | x2 2N | K1 16 | K2 16 | samples (5S int32) ] in,
    //                              [ FE 180 S | cost 10 S | count (10 S int32) ] scratch, [ result | mask ] out
    const size_t in_d = (size_t)4 * N + 32 + ((size_t)5 * S + 1) / 2;
    const size_t scr_d = (size_t)180 * S + (size_t)10 * S + ((size_t)10 * S + 1) / 2;
    const size_t res_d = (epi_result_bytes() + 7) / 8;
    const size_t out_d = res_d + ((size_t)N + 7) / 8;
    int rc = ensure_pnp(ctx, in_d + scr_d + out_d + 8);
    if (rc != CLC_OK) return rc;
    rc = ensure_pinned(ctx, (in_d > out_d ? in_d : out_d) * sizeof(double) + 64);
    if (rc != CLC_OK) return rc;
    double* d1 = ctx->d_pnp;
    double* d2 = d1 + (size_t)2 * N;
    double* dK1 = d2 + (size_t)2 * N;
    double* dK2 = dK1 + 16;
    int32_t* dSamples = (int32_t*)(dK2 + 16);
    double* dFE = ctx->d_pnp + in_d;
    double* dCost = dFE + (size_t)180 * S;
    int32_t* dCount = (int32_t*)(dCost + (size_t)10 * S);
    double* dRes = ctx->d_pnp + in_d + scr_d;
    uint8_t* dMask = (uint8_t*)(dRes + res_d);
    double* hp = (double*)ctx->h_pin;
    memcpy(hp, h_x1, sizeof(double) * 2 * N);
    memcpy(hp + (size_t)2 * N, h_x2, sizeof(double) * 2 * N);
    memcpy(hp + (size_t)4 * N, h_K1, sizeof(double) * 9);
    memcpy(hp + (size_t)4 * N + 16, h_K2, sizeof(double) * 9);
    memcpy(hp + (size_t)4 * N + 32, h_samples, sizeof(int32_t) * 5 * S);
    CLC_HIP(ctx, hipMemcpyAsync(d1, hp, in_d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, launch_essential_ransac(d1, d2, N, dK1, dK2, dSamples, S, thr2, dFE, dCount, dCost, dMask, dRes, ctx->stream, &ctx->prof));
    CLC_HIP(ctx, hipMemcpyAsync(hp, dRes, out_d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<double> fe;
    if (h_all_E) { fe.resize((size_t)180 * S); CLC_HIP(ctx, hipMemcpyAsync(fe.data(), dFE, sizeof(double) * 180 * S, hipMemcpyDeviceToHost, ctx->stream)); }
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    struct { double E[9]; double F[9]; double cost; int32_t h; int32_t count; } r;
    memcpy(&r, hp, sizeof r);
    if (h_E) memcpy(h_E, r.E, sizeof r.E);
    if (h_F) memcpy(h_F, r.F, sizeof r.F);
    if (h_mask) memcpy(h_mask, (const uint8_t*)(hp + res_d), (size_t)N);
    if (n_inliers) *n_inliers = r.h >= 0 ? r.count : 0;
    if (h_all_E)
        for (size_t k = 0; k < (size_t)10 * S; ++k) memcpy(h_all_E + 9 * k, fe.data() + 18 * k + 9, sizeof(double) * 9);
    return CLC_OK;
}

int clc_essential_ransac(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1, const double* h_K2,
                         const int32_t* h_samples, int S, uint64_t seed, double thr2, double* h_E, double* h_F,
                         uint8_t* h_inlier_mask, int* n_inliers)
{
    return essential_impl(ctx, h_x1, h_x2, N, h_K1, h_K2, h_samples, S, seed, thr2, h_E, h_F, h_inlier_mask, n_inliers, nullptr);
}

int clc_essential_fivepoint(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1, const double* h_K2,
                            const int32_t* h_samples, int S, double* h_E_out)
{
    if (!h_samples || !h_E_out) return fail(ctx, CLC_ERR_BAD_ARG, "essential_fivepoint: bad argument");
    return essential_impl(ctx, h_x1, h_x2, N, h_K1, h_K2, h_samples, S, 0, 1.0, nullptr, nullptr, nullptr, nullptr, h_E_out);
}

int clc_pnp_ransac(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const int32_t* h_samples,
                   int S, uint64_t seed, double thr2, double* h_Rt, uint8_t* h_inlier_mask, int* n_inliers, double* cost)
{
    return pnp_ransac_impl(ctx, h_X, h_x, N, h_K, h_samples, S, seed, thr2, h_Rt, h_inlier_mask, n_inliers, cost, nullptr);
}

int clc_pnp_refine(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const uint8_t* h_inlier_mask,
                   const double* h_Rt_in, double huber_a, int max_iter, double* h_Rt_out, double* h_cov, double* rmse, int* iterations)
{
    if (!ctx || N < 0 || !h_K || !h_Rt_in || (N > 0 && (!h_X || !h_x))) return fail(ctx, CLC_ERR_BAD_ARG, "pnp_refine: bad argument");
    if (N < 3) return fail(ctx, CLC_ERR_BAD_ARG, "pnp_refine: needs at least 3 correspondences");
    if (!(huber_a > 0.0)) huber_a = 16.0;          // ceres::HuberLoss(Square(4.0)), Refiner.hpp:122
    if (max_iter <= 0) max_iter = 50;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    // workspace in doubles: [ X 3N | x 2N | K 16 | Rt_in 12 | mask (N bytes) ] in, [ RefineOut ] out
    const size_t mask_d = ((size_t)N + 7) / 8;
    const size_t in_d = (size_t)5 * N + 16 + 12 + mask_d;
    const size_t out_d = (pnp_refine_out_bytes() + 7) / 8;
    int rc = ensure_pnp(ctx, in_d + out_d + 8);
    if (rc != CLC_OK) return rc;
    rc = ensure_pinned(ctx, (in_d > out_d ? in_d : out_d) * sizeof(double) + 64);
    if (rc != CLC_OK) return rc;
    double* dX = ctx->d_pnp;
    double* dx = dX + (size_t)3 * N;
    double* dK = dx + (size_t)2 * N;
    double* dRt = dK + 16;
    uint8_t* dMask = (uint8_t*)(dRt + 12);
    double* dOut = ctx->d_pnp + in_d;
    double* hp = (double*)ctx->h_pin;
    memcpy(hp, h_X, sizeof(double) * 3 * N);
    memcpy(hp + (size_t)3 * N, h_x, sizeof(double) * 2 * N);
    memcpy(hp + (size_t)5 * N, h_K, sizeof(double) * 9);
    memcpy(hp + (size_t)5 * N + 16, h_Rt_in, sizeof(double) * 12);
    if (h_inlier_mask) memcpy(hp + (size_t)5 * N + 28, h_inlier_mask, (size_t)N);
    CLC_HIP(ctx, hipMemcpyAsync(dX, hp, in_d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, launch_pnp_refine(dRt, dX, dx, h_inlier_mask ? dMask : nullptr, N, dK, huber_a, max_iter, dOut, ctx->stream, &ctx->prof));
    CLC_HIP(ctx, hipMemcpyAsync(hp, dOut, out_d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    struct { double Rt[12]; double cov[36]; double cost; double rmse; int32_t iterations; int32_t n_used; } r;
    memcpy(&r, hp, sizeof r);
    if (h_Rt_out) memcpy(h_Rt_out, r.Rt, sizeof r.Rt);
    if (h_cov) memcpy(h_cov, r.cov, sizeof r.cov);
    if (rmse) *rmse = r.rmse;
    if (iterations) *iterations = r.iterations;
    return CLC_OK;
}

int clc_pnp_localize(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const int32_t* h_samples,
                     int S, uint64_t seed, double thr2, double huber_a, double* h_Rt, double* h_cov, uint8_t* h_inlier_mask,
                     int* n_inliers, double* rmse)
{
    if (h_Rt) memset(h_Rt, 0, sizeof(double) * 12);
    if (h_cov) memset(h_cov, 0, sizeof(double) * 36);
    if (rmse) *rmse = 0.0;
    return pnp_ransac_impl(ctx, h_X, h_x, N, h_K, h_samples, S, seed, thr2, h_Rt, h_inlier_mask, n_inliers, nullptr, nullptr,
                           huber_a > 0.0 ? huber_a : 16.0, h_cov, rmse);
}

int clc_pnp_p3p(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const int32_t* h_samples, int S,
                double* h_Rt_out)
{
    if (!h_samples || !h_Rt_out) return fail(ctx, CLC_ERR_BAD_ARG, "pnp_p3p: bad argument");
    return pnp_ransac_impl(ctx, h_X, h_x, N, h_K, h_samples, S, 0, 1.0, nullptr, nullptr, nullptr, nullptr, h_Rt_out);
}

} // extern "C"
