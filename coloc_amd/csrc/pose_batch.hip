// pose_batch.hip -- the a-contrario robust solvers behind the C ABI (include/coloc_hip.h): what Localizer::localizeImage
// (reference include/coloc/Localizer.hpp:82-93, SfM_Localizer::Localize with error_max = +inf) and RobustMatcher::filterEssential
// (RobustMatcher.hpp:153-171, robust::ACRANSAC) run -- and, since round 6, filterFundamental / filterHomography (:128-151, :188-239) --,
// single solves and batches of them driven from one host thread.
#include "clc_ctx.h"
#include "clc_acr.h"
#include "twoview_min.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace clc;

namespace {

// log10 C(n, k) and log10 C(k, m), k = 0..n, as OpenMVG tabulates them: a FLOAT log10 table, float accumulation
// (logcombi<float>); the O(n^2) re-summation of the prefix is replaced by the running prefix -- same additions, same order.
// `lg` = the context's table of (float) log10(k), grown on demand and kept: it does not depend on n, and n + 1 calls of the
// portable log10 per solve were 40 us of a 220 us solve at n = 1000 (200 us at n = 5000).
void acr_tables(int n, int m, float* logc_n, float* logc_k, std::vector<float>& lg)
{
    if (lg.size() < (size_t)n + 2) {
        const size_t have = lg.size() < 1 ? 1 : lg.size();
        lg.resize((size_t)n + 2, 0.0f);
        lg[0] = 0.0f;
        for (size_t k = have; k < lg.size(); ++k) lg[k] = (float)clc_acr_log10((double)k);
    }
    std::vector<float> prefix((size_t)n + 1, 0.0f);
    for (int i = 1; i <= n; ++i) prefix[i] = prefix[i - 1] + (lg[n - i + 1] - lg[i]);
    for (int k = 0; k <= n; ++k) {
        uint32_t kk = (uint32_t)k;
        if (kk >= (uint32_t)n || kk == 0) { logc_n[k] = 0.0f; continue; }
        if ((uint32_t)n - kk < kk) kk = (uint32_t)n - kk;
        logc_n[k] = prefix[kk];
    }
    for (int nn = 0; nn <= n; ++nn) {
        uint32_t kk = (uint32_t)m;
        float r = 0.0f;
        if (!(kk >= (uint32_t)nn || kk == 0)) {
            if ((uint32_t)nn - kk < kk) kk = (uint32_t)nn - kk;
            for (uint32_t i = 1; i <= kk; ++i) r += lg[nn - i + 1] - lg[i];
        }
        logc_k[nn] = r;
    }
}

size_t dbl(size_t bytes) { return (bytes + 7) / 8; }

// kind 0: a = X (3 N), b = x (2 N), K1 = intrinsics; kind 1: a = x1, b = x2 (2 N each), K1 / K2, image 2 of img_w x img_h;
// kinds 2 / 3 (round 6: RobustMatcher's 'F' / 'H' models, RobustMatcher.hpp:128-151, :188-239): a = x1, b = x2 in pixels, both images
// img_w x img_h -- the points are conditioned by the image size on their way into the pinned block (ACKernelAdaptor), the rounds are
// the resection's one-launch rounds with the seven-point / four-point solve in place of P3P, the model is brought back to pixels in finish().
// h_model: 12 doubles [R|t] (kind 0), {E (9), F (9)} (kind 1), F or H (9, pixels; kinds 2 / 3).
//
// One a-contrario solve as a small state machine (round 4): begin() stages the inputs and enqueues the first two rounds, poll() looks at the
// pinned progress word ONCE -- if the round the host waits for has come out it enqueues the next one (rounds stay enqueued one ahead of
// what the host knows) or moves on to the refinement, whose record it then polls the same way --, finish() copies the result out.  A
// single solve spins on poll() exactly as the loop it replaces did; clc_pnp_localize_ac_batch drives SEVERAL solves, each on a context of
// its own, from one thread: a solve is a chain of short launches with the host in the loop and leaves the GPU idle most of the time, so
// the chains of independent cameras interleave (BASELINE config[2]: "batched PnP/RANSAC pose").
struct AcrRun {
    // arguments
    clc_ctx* ctx = nullptr;
    int kind = 0, N = 0, img_w = 0, img_h = 0, max_iteration = 0;
    const double *h_a = nullptr, *h_b = nullptr, *h_K1 = nullptr, *h_K2 = nullptr;
    uint64_t seed = 0;
    double precision = 0.0, refine_huber = -1.0;
    double* h_model = nullptr; uint8_t* h_mask = nullptr; int32_t* h_inliers = nullptr;
    int *n_inliers = nullptr, *iterations = nullptr, *rounds = nullptr;
    double *error_max = nullptr, *min_nfa = nullptr, *h_cov = nullptr, *rmse = nullptr;
    // state
    enum Phase { IDLE, ROUNDS, REFINE, DONE } phase = IDLE;
    int status = CLC_OK;
    int m = 0, M = 0, md = 0;
    bool refine = false;
    AcrProblem pb{};
    tv::Normalizer norm_t{};          // kinds 2, 3: the conditioning of both point sets
    hipStream_t st = nullptr;
    int launches = 0, bound = 0, reserve0 = 0;
    uint32_t round = 0, spins = 0;
    std::chrono::steady_clock::time_point wait_start;
    // lockstep batches (drive_group): the solve's rounds ride in launches shared with the other solves of the batch, on group_stream; the
    // run only watches its word -- `reported` = the round waited for has come out, `more` = it needs another one (bound: its batch bound)
    bool grouped = false, reported = false, more = false;
    hipStream_t group_stream = nullptr, refine_st = nullptr;
    int first_bound = 0;
    const double* stage_src = nullptr; double* stage_dst = nullptr; size_t stage_n = 0;   // the inputs' way to the device (a launch, not a copy)
    int batch_cap = kAcrMaxBatch;      // most iterations a round evaluates (CLC_ACR_BATCH_CAP; batches: see acr_batch_cap)
    unsigned long long* h_word = nullptr;
    AcrResult* h_res = nullptr;
    int32_t* p_inl = nullptr;
    double* p_ref = nullptr;
    int32_t* ready = nullptr;
    double *d_a = nullptr, *d_b = nullptr, *d_K1 = nullptr, *d_K2 = nullptr, *d_models = nullptr, *d_ref = nullptr;
    AcrState* d_state = nullptr; AcrHyp* d_hyp = nullptr;
    uint32_t *d_sorted = nullptr, *d_best = nullptr, *d_index = nullptr;
    AcrResult* d_res = nullptr; uint8_t* d_mask = nullptr;

    // Every failure after the first launch drains the stream first (ignoring what the drain reports): launches of the failed solve may
    // still be in flight and would otherwise write the progress word / result record of the NEXT solve, which reuses the same pinned block.
    int drained(const int code) { (void)hipStreamSynchronize(st); phase = DONE; status = code; return code; }
    int stop(const int code) { phase = DONE; status = code; return code; }

    int enqueue_round(const int bnd)
    {
        const int S = bnd < 1 ? 1 : (bnd > kAcrMaxBatch ? kAcrMaxBatch : bnd);
        if (kind != 1) {
            // one launch: replay of the previous round, this round's samples, P3P (seven-point, four-point), residuals / sort / NFA;
            // the word of round r comes out of launch r + 1
            CLC_HIP(ctx, launch_acr_round_p3p(pb, launches & 1, d_state, d_hyp, d_sorted, d_models, d_best, d_index, h_word, st, S, d_mask,
                                              d_res, nullptr, p_inl, h_res));
        } else {
            // two launches: replay of the previous round + this round's samples + five-point solve, then nfa; the word of round r
            // comes out of round r + 1's first launch
            CLC_HIP(ctx, launch_acr_round_5pt(pb, launches & 1, d_state, d_hyp, d_sorted, d_models, d_best, d_index, h_word, st, S, d_mask,
                                              d_res, nullptr, p_inl, h_res));
        }
        ++launches;
        return CLC_OK;
    }

    // validates, stages, enqueues the first two rounds.  Returns a status; phase == DONE afterwards means there is nothing to wait for.
    int begin()
    {
        static const int k_m[4] = { 3, 5, 7, 4 }, k_M[4] = { 4, 10, 3, 1 }, k_md[4] = { 12, 18, 12, 12 };
        if (kind < 0 || kind > 3) return stop(fail(ctx, CLC_ERR_BAD_ARG, "acransac: unknown model"));
        m = k_m[kind]; M = k_M[kind]; md = k_md[kind];
        const int ad = kind == 0 ? 3 : 2;
        if (!ctx || N < 0 || max_iteration < 0 || (kind <= 1 && !h_K1) || (kind == 1 && !h_K2) || (N > 0 && (!h_a || !h_b)))
            return stop(fail(ctx, CLC_ERR_BAD_ARG, "acransac: bad argument"));
        if (n_inliers) *n_inliers = 0;
        if (error_max) *error_max = 0.0;
        if (min_nfa) *min_nfa = INFINITY;
        if (iterations) *iterations = 0;
        if (rounds) *rounds = 0;
        if (h_mask && N > 0) memset(h_mask, 0, (size_t)N);
        if (N <= m || max_iteration == 0) return stop(CLC_OK);                 // ACRANSAC: nData <= sizeSample -> (0, 0), no model
        if (N > kAcrMaxN) return stop(fail(ctx, CLC_ERR_CAPACITY, "acransac: more than 16384 correspondences per solve"));
        if (max_iteration > 500000) return stop(fail(ctx, CLC_ERR_CAPACITY, "acransac: more than 500000 iterations"));
        if (kind >= 1 && (img_w <= 0 || img_h <= 0)) return stop(fail(ctx, CLC_ERR_BAD_ARG, "acransac: image size needed for the two-view models"));
        phase = DONE; status = CLC_ERR_HIP;                                     // (what an early CLC_HIP return leaves behind)
        const int rc0 = begin_body(ad);
        if (rc0 != CLC_OK) { phase = DONE; status = rc0; }
        return rc0;
    }

    int begin_body(const int ad)
    {
        CLC_HIP(ctx, hipSetDevice(ctx->device));
        refine = kind == 0 && refine_huber > 0.0;
        // device workspace (doubles): [ a | b | K1 16 | K2 16 | logc_n | logc_k | initial state ] staged by
        // one launch, then scratch
        const size_t state_d = 2 * dbl(sizeof(AcrState));                      // [ copy 0 | copy 1 = the state a run starts from ]
        const size_t in_d = (size_t)(ad + 2) * N + 32 + 2 * dbl(sizeof(float) * ((size_t)N + 1)) + state_d;
        // a round's launches read what the round before them wrote: two copies of state, models, slots and sorted lists, indexed by
        // launch parity (acransac.hip: acr_round_kernel, acr_solve5_kernel)
        if (grouped) batch_cap = kind == 1 ? 12 : 8;                   // (a shared launch carries every chain's speculative slots: shorter rounds)
        if (const char* e = getenv("CLC_ACR_BATCH_CAP")) { const int v = atoi(e); if (v >= 1 && v <= kAcrMaxBatch) batch_cap = v; }
        const int copies = 2;
        const int Ms = kind == 1 ? 10 : 4;                             // slots per iteration between the parity copies (the kernels' stride)
        const size_t models_d = (size_t)copies * kAcrMaxBatch * Ms * md;
        const size_t hyp_d = dbl(acr_hyp_bytes() * copies * kAcrMaxBatch * Ms);
        const size_t sorted_d = dbl(sizeof(uint32_t) * (size_t)copies * kAcrMaxBatch * Ms * N);
        const size_t idx_d = dbl(sizeof(uint32_t) * (size_t)N);
        const size_t res_d = dbl(sizeof(AcrResult)), mask_d = dbl((size_t)N);
        const size_t ref_d = refine ? dbl(pnp_refine_out_bytes()) : 0;
        int rc = ensure_pnp(ctx, in_d + models_d + hyp_d + sorted_d + 2 * idx_d + res_d + mask_d + ref_d + 16);
        if (rc != CLC_OK) return rc;
        // pinned: [ inputs | state mirror | sequence word | result | mask | inlier list | refine record ]
        const size_t inl_d = dbl(sizeof(int32_t) * (size_t)N);
        rc = ensure_pinned(ctx, (in_d + state_d + 1 + res_d + mask_d + inl_d + ref_d) * sizeof(double) + 64);   // (+1: the polled word)
        if (rc != CLC_OK) return rc;
        double* d = ctx->d_pnp;
        d_a = d;                               d += (size_t)ad * N;
        d_b = d;                               d += (size_t)2 * N;
        d_K1 = d;                              d += 16;
        d_K2 = d;                              d += 16;
        float* d_cn = (float*)d;               d += dbl(sizeof(float) * ((size_t)N + 1));
        float* d_ck = (float*)d;               d += dbl(sizeof(float) * ((size_t)N + 1));
        d_state = (AcrState*)d;                d += state_d;
        d_models = d;                          d += models_d;
        d_hyp = (AcrHyp*)d;                    d += hyp_d;
        d_sorted = (uint32_t*)d;               d += sorted_d;
        d_best = (uint32_t*)d;                 d += idx_d;
        d_index = (uint32_t*)d;                d += idx_d;
        d_res = (AcrResult*)d;                 d += res_d;
        d_mask = (uint8_t*)d;                  d += mask_d;
        d_ref = d;
        double* hp = (double*)ctx->h_pin;
        if (kind >= 2) {
            // ACKernelAdaptor: NormalizePoints(x, &x_, &N_, w, h) for both point sets (x_n = d x + t, oracle/clc_oracle_twoview.c orc_tv_normalize)
            norm_t = tv::normalizer(img_w, img_h);
            double* q1 = hp;
            double* q2 = hp + (size_t)2 * N;
            for (int i = 0; i < N; ++i) {
                q1[2 * i] = h_a[2 * i] * norm_t.d + norm_t.tx; q1[2 * i + 1] = h_a[2 * i + 1] * norm_t.d + norm_t.ty;
                q2[2 * i] = h_b[2 * i] * norm_t.d + norm_t.tx; q2[2 * i + 1] = h_b[2 * i + 1] * norm_t.d + norm_t.ty;
            }
        } else {
            memcpy(hp, h_a, sizeof(double) * ad * N);
            memcpy(hp + (size_t)ad * N, h_b, sizeof(double) * 2 * N);
        }
        double* hK = hp + (size_t)(ad + 2) * N;
        memset(hK, 0, sizeof(double) * 32);
        if (h_K1) memcpy(hK, h_K1, sizeof(double) * 9);
        if (h_K2) memcpy(hK + 16, h_K2, sizeof(double) * 9);
        float* h_cn = (float*)(hK + 32);
        float* h_ck = (float*)(hK + 32 + dbl(sizeof(float) * ((size_t)N + 1)));
        acr_tables(N, m, h_cn, h_ck, ctx->acr_lg);
        // the state ACRANSAC starts from is part of the upload (every round draws its own samples on the device)
        AcrState* h_states = (AcrState*)(hK + 32 + 2 * dbl(sizeof(float) * ((size_t)N + 1)));
        memset(h_states, 0, 2 * sizeof(AcrState));
        // launch 0 (parity 0) reads copy 1
        AcrState* h_init = h_states + 1;
        h_init->min_nfa = INFINITY; h_init->error_max = INFINITY;
        h_init->best_iter = -1;
        h_init->reserve = max_iteration / 10;
        h_init->n_iter = max_iteration - h_init->reserve;
        h_init->n_index = N; h_init->index_all = 1;
        h_init->ac_mode = std::isinf(precision) ? 1 : 0;
        h_init->grow = batch_cap < 32 ? batch_cap : 32;
        h_init->cur_batch = h_init->n_iter < h_init->grow ? h_init->n_iter : h_init->grow;
        h_word = (unsigned long long*)(hp + in_d + state_d);
        h_res = (AcrResult*)(hp + in_d + state_d + 1);
        p_inl = (int32_t*)(hp + in_d + state_d + 1 + res_d + mask_d);
        p_ref = hp + in_d + state_d + 1 + res_d + mask_d + inl_d;
        __atomic_store_n(h_word, 0ull, __ATOMIC_RELAXED);

        pb = AcrProblem{};
        pb.kind = kind; pb.n = N; pb.m = m; pb.max_models = M; pb.model_doubles = md; pb.batch_cap = batch_cap;
        pb.a = d_a; pb.b = d_b; pb.K1 = d_K1; pb.K2 = d_K2; pb.logc_n = d_cn; pb.logc_k = d_ck;
        pb.loge0 = clc_acr_log10((double)M * (double)(N - m));
        if (kind == 0) {
            // ACKernelAdaptorResection_Intrinsics: residuals on the normalised camera plane (x 1 / focal), logalpha0 = log10(pi)
            pb.logalpha0 = clc_acr_log10(M_PI);
            pb.mult = 1.0;
            pb.norm = 1.0 / h_K1[0];
            for (int e = 0; e < 9; ++e) pb.K1v[e] = h_K1[e];
        } else if (kind == 1) {
            // ACKernelAdaptorEssential: point-to-line, logalpha0 = log10(2 D / A * 0.5) of image 2, error^(1/2)
            const double D = sqrt((double)img_w * (double)img_w + (double)img_h * (double)img_h), A = (double)img_w * (double)img_h;
            pb.logalpha0 = clc_acr_log10(2.0 * D / A * .5);
            pb.mult = 0.5;
            pb.norm = 1.0;
        } else {
            // ACKernelAdaptor on conditioned points: 'F' point-to-line, log10(2 D / A / N2(0,0)), error^(1/2); 'H' point-to-point,
            // log10(pi / A / N2(0,0)^2); thresholds and the reported precision scale with N2(0,0) = d
            const double D = sqrt((double)img_w * (double)img_w + (double)img_h * (double)img_h), A = (double)img_w * (double)img_h;
            pb.norm = norm_t.d;
            if (kind == 2) { pb.logalpha0 = clc_acr_log10(2.0 * D / A / pb.norm); pb.mult = 0.5; }
            else { pb.logalpha0 = clc_acr_log10(M_PI / A / (pb.norm * pb.norm)); pb.mult = 1.0; }
        }
        pb.max_threshold = std::isinf(precision) ? INFINITY : precision * (pb.norm * pb.norm);
        pb.seed = seed;

        st = grouped ? group_stream : ctx->stream;
        stage_src = hp; stage_dst = ctx->d_pnp; stage_n = (in_d + 1) & ~(size_t)1;      // (both blocks are sized past in_d + 1)
        if (!grouped) CLC_HIP(ctx, launch_acr_stage(stage_src, stage_dst, stage_n, st));  // (grouped: one launch for the batch, drive_group)
        if (!grouped) prof_mark(&ctx->prof, CLC_KERNEL_PNP_SCORE, true, st);
        // Rounds are enqueued ONE AHEAD of what the host knows: the solve / nfa / select kernels take the round's batch from the
        // device state (a round enqueued after the run has finished finds nothing to do), so the GPU goes from one round's
        // select straight into the next round's solve while the host is still polling (a 10 us bubble per round otherwise).
        launches = 0;
        // Upper bound of the batch a round can ask for, from what the host knows when it enqueues it (one or two rounds behind the
        // device): while the index set has not switched the batch doubles up to kAcrMaxBatch; afterwards it is what is left of the
        // reserve, remaining = n_iter - iter (+ a margin for the "no inliers: n_iter++" rule, once per round).
        reserve0 = h_init->reserve;
        bound = h_init->n_iter < batch_cap ? h_init->n_iter : batch_cap;
        first_bound = h_init->cur_batch;                               // (a replaying launch takes its first batch as it stands)
        if (!grouped) {
            int rc2 = enqueue_round(first_bound);
            if (rc2 != CLC_OK) return drained(rc2);
            rc2 = enqueue_round(bound);                                 // speculative: the round after the one being waited for
            if (rc2 != CLC_OK) return drained(rc2);
        }                                                               // (grouped: drive_group enqueues the shared launches)
        round = 1;
        spins = 0;
        reported = false; more = false;
        wait_start = std::chrono::steady_clock::now();
        phase = ROUNDS;
        status = CLC_OK;
        return CLC_OK;
    }

    // this solve's part of a shared launch
    void chain(AcrChain& c) const
    {
        c.pb = pb;
        c.states = d_state; c.hyps = d_hyp; c.sorted = d_sorted; c.models = d_models; c.best_inliers = d_best; c.index_set = d_index;
        c.h_word = h_word;
        c.fin = AcrFinish{ d_mask, d_res, nullptr, p_inl, h_res };
    }
    // the shared launch of the next round is in the stream: wait for that round's word
    void advance()
    {
        ++round; spins = 0; reported = false; more = false;
        wait_start = std::chrono::steady_clock::now();
    }

    // One look at the progress word / the refinement's ready flag.  Returns the status; phase == DONE when the solve has ended.
    int poll()
    {
        if (phase == ROUNDS) {
            // the select kernel publishes one packed word (round number, iterations consumed, iter, n_iter) in pinned memory: poll it
            // (a stream synchronisation costs ~10 us per round); after 2 ms without progress fall back to the synchronisation, which
            // also surfaces errors
            unsigned long long w = __atomic_load_n(h_word, __ATOMIC_ACQUIRE);
            if ((w >> 49) < (round & 0x7FFFu)) {
                if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - wait_start > std::chrono::milliseconds(2)) {
                    const hipError_t e = hipStreamSynchronize(st);
                    if (e != hipSuccess) return drained(fail(ctx, CLC_ERR_HIP, "hipStreamSynchronize(st)", e));
                    w = __atomic_load_n(h_word, __ATOMIC_ACQUIRE);
                    if ((w >> 49) < (round & 0x7FFFu)) return drained(fail(ctx, CLC_ERR_HIP, "acransac: round did not complete"));
                } else return CLC_OK;
            }
            const int iter_k = (int)(w & 0xFFFFFu), n_iter_k = (int)((w >> 20) & 0xFFFFFu);
            if (iter_k < n_iter_k) {
                const bool switched = ((w >> 48) & 1u) != 0;
                const long left = (long)n_iter_k - iter_k + 4 + (switched ? 0 : reserve0);
                bound = left > batch_cap ? batch_cap : (int)left;
                if (round > 0x7000u) return drained(fail(ctx, CLC_ERR_STATE, "acransac: too many rounds"));
                if (grouped) { reported = true; more = true; return CLC_OK; }      // (drive_group enqueues the batch's next launch)
                const int rc = enqueue_round(bound);                   // speculative: the round after the one being waited for
                if (rc != CLC_OK) return drained(rc);
                ++round;
                spins = 0;
                wait_start = std::chrono::steady_clock::now();
                return CLC_OK;
            }
            // done: the completing round has left the result in pinned memory
            reported = true; more = false;
            if (!grouped) prof_mark(&ctx->prof, CLC_KERNEL_PNP_SCORE, false, st);
            // The result record, mask and inlier list were written by the round that completed the run BEFORE its word (system-scope
            // release / acquire): no finish launch, and without refinement no stream synchronisation either -- the round enqueued ahead is
            // still in the stream, evaluates nothing and touches no host memory; anything enqueued later on this stream is ordered behind it.
            if (!refine) { phase = DONE; return CLC_OK; }
            // Behind the launch that completed the run, on the same stream (nothing is queued behind that launch any more: the word of
            // the last round comes out of the last launch).  The refinement writes its record into pinned memory and sets `ready` last;
            // the host polls that instead of synchronising the stream (~5 us), with the synchronisation as the fallback after 5 ms.
            // (A grouped run refines on its OWN context's stream: the shared stream still carries the other solves' rounds.  What the
            // refinement reads was written before the word the host has just seen -- system-scope release / acquire -- so the launch
            // needs no ordering against the shared stream.)
            ready = (int32_t*)((uint8_t*)p_ref + pnp_refine_ready_offset());
            __atomic_store_n(ready, 0, __ATOMIC_RELAXED);
            refine_st = grouped ? ctx->stream : st;
            const hipError_t e = launch_pnp_refine((const double*)d_res /* AcrResult.model = [R|t] */, d_a, d_b, d_mask, N, d_K1, refine_huber, 50,
                                                   d_ref, refine_st, &ctx->prof, &d_res->valid, p_ref);
            if (e != hipSuccess) return drained(fail(ctx, CLC_ERR_HIP, "launch_pnp_refine", e));
            spins = 0;
            wait_start = std::chrono::steady_clock::now();
            phase = REFINE;
            return CLC_OK;
        }
        if (phase == REFINE) {
            if (__atomic_load_n(ready, __ATOMIC_ACQUIRE) == 0) {
                if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - wait_start > std::chrono::milliseconds(5)) {
                    const hipError_t e = hipStreamSynchronize(refine_st);
                    if (e != hipSuccess) return drained(fail(ctx, CLC_ERR_HIP, "hipStreamSynchronize(refine stream)", e));
                    if (__atomic_load_n(ready, __ATOMIC_ACQUIRE) == 0) return drained(fail(ctx, CLC_ERR_HIP, "acransac: refinement did not complete"));
                } else return CLC_OK;
            }
            phase = DONE;
        }
        return status;
    }

    // after phase == DONE with status CLC_OK and a run that was started: the result into the caller's buffers
    void finish()
    {
        if (status != CLC_OK || !h_res) return;
        const AcrResult r = *h_res;
        if (h_model) {
            if (kind == 0) memcpy(h_model, r.model, sizeof(double) * 12);
            else if (kind >= 2) {
                // Unnormalize(&model): F = N2^T Fn N1, H = N2^-1 Hn N1 (no model: zeros stay zeros)
                if (r.n_inliers > 0) tv::unnormalize(kind == 3, norm_t, r.model, h_model);
                else memset(h_model, 0, sizeof(double) * 9);
            } else { memcpy(h_model, r.model + 9, sizeof(double) * 9); memcpy(h_model + 9, r.model, sizeof(double) * 9); }   // slots hold {F, E}
        }
        // the mask is rebuilt from the inlier list here (h_mask was cleared above): the device does not push N bytes + one scattered byte
        // per inlier over PCIe for it
        if (h_mask) for (int i = 0; i < r.n_inliers; ++i) h_mask[p_inl[i]] = 1;
        if (h_inliers && r.n_inliers > 0) memcpy(h_inliers, p_inl, sizeof(int32_t) * (size_t)r.n_inliers);
        if (n_inliers) *n_inliers = r.n_inliers;
        if (error_max) *error_max = r.error_max;
        if (min_nfa) *min_nfa = r.min_nfa;
        if (iterations) *iterations = r.iterations;
        if (rounds) *rounds = r.rounds;
        if (refine) {
            struct { double Rt[12]; double cov[36]; double cost; double rmse; int32_t iterations; int32_t n_used; } f;
            memcpy(&f, p_ref, sizeof f);
            if (r.n_inliers > 0) {
                if (h_model) memcpy(h_model, f.Rt, sizeof f.Rt);
                if (h_cov) memcpy(h_cov, f.cov, sizeof f.cov);
                if (rmse) *rmse = f.rmse;
            }
        }
    }
};

// All runs to their end from ONE host thread: whichever solve's round has come out gets its next one enqueued (a run is a chain of short
// launches with the host in the loop, so several runs interleave on the device).  Round 5 measured two and three driving threads (the
// runs dealt out, each thread on its runs' own contexts and streams): eight two-view filters 0.134 -> 0.136 ms per pair, eight poses
// 0.049 -> 0.052 ms per pose -- the chains' own latency, not the host's launch calls, is what a batch waits for; one thread stays.
void drive_runs(std::vector<AcrRun>& runs)
{
    size_t live = 0;
    for (AcrRun& r : runs) if (r.phase != AcrRun::DONE) ++live;
    while (live > 0)
        for (AcrRun& r : runs) {
            if (r.phase == AcrRun::DONE) continue;
            (void)r.poll();
            if (r.phase == AcrRun::DONE) --live;
        }
}

// Lockstep form of the same (round 5, the default of the batched entries): the batch's solves -- one kind, contexts on one device -- share
// their launches.  Round r of every unfinished solve is ONE launch (resection) or two (two-view) with blockIdx.y = solve
// (launch_acr_round_*_chains), on the first context's stream, enqueued one ahead as for a single solve; the host waits for the round's
// words of all unfinished solves and enqueues the next shared launch while any of them needs one.  A finished solve's part of the later
// launches finds nothing to replay and returns.  Why: eight interleaved poses were ~80 launches from one thread (4-5 us each inside the
// runtime, and the runtime serialises launching threads), eight two-view filters ~100; in lockstep they are ~10 and ~16.  Same bits per
// solve: a chain's kernels take its batch from its own device state, the grid and the sort width (the largest chain's) only bound them.
// Measured (MI355X, N = 1 000, 30 % outliers, tools/time_two_view.py; interleaved -> lockstep, one staging launch
// for the batch, a round's launches carrying only the solves still in their rounds):
//     two-view filters   2: 0.417 -> 0.439 ms   4: 0.570 -> 0.574   8: 1.08-1.21 -> 0.75-0.80      (rounds of <= 12 / 16 iterations)
//     resection poses    2: 0.190 -> 0.189      4: 0.251 -> 0.271   8: 0.514 -> 0.508; with rounds of <= 8 iterations 0.429 (4: 0.295)
// The interleaved batch is bound by the host's launch calls (eight poses = ~70 launches of 5-7 us from one thread, two of them in flight on
// the device at a time); in lockstep eight poses are ~12 launches, but each carries every solve's speculative slots -- 8 x 32 iterations x 4
// models of 1 024 threads do not fit the chip at once (33 us per round against 17.5) -- hence the cap on a round's iterations, which costs
// rounds.  The two-view round is a 44 us chain of dependent fp64 steps in ONE wave per iteration: eight chains' solves in one launch cost
// what one costs.  Default: lockstep for two-view batches of four or more (rounds of <= 12 iterations) and resection batches of eight or
// more (<= 8); CLC_ACR_LOCKSTEP=1 / =0 forces it on (any batch of two or more) / off for both kinds, CLC_ACR_BATCH_CAP the iterations.
bool acr_lockstep(const int kind, const int n_jobs)
{
    static const int mode = [] { const char* e = getenv("CLC_ACR_LOCKSTEP"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
    if (n_jobs < 2 || mode == 0) return false;
    if (mode == 1) return true;
    return kind == 1 ? n_jobs >= 4 : n_jobs >= 8;      // (kinds 2, 3 run the resection's one-launch rounds: its break-even)
}
void drive_group(std::vector<AcrRun>& runs)
{
    std::vector<AcrRun*> live;
    for (AcrRun& r : runs) if (r.phase == AcrRun::ROUNDS && r.grouped) live.push_back(&r);
    if (live.empty()) { drive_runs(runs); return; }
    const int kind = live[0]->kind;
    hipStream_t st = live[0]->group_stream;
    clc_ctx* ctx0 = live[0]->ctx;
    int launches = 0;
    auto fail_all = [&](const int code) { for (AcrRun* r : live) if (r->phase != AcrRun::DONE) (void)r->drained(code); };
    if (hipSetDevice(ctx0->device) != hipSuccess) { fail_all(fail(ctx0, CLC_ERR_HIP, "hipSetDevice")); return; }
    // A solve's workspace (its context's d_pnp: inputs, both state copies, slots) may still be written by launches that an EARLIER solve
    // left on that context's OWN stream -- a single or interleaved solve returns when its word says "done", the round enqueued ahead of
    // it is still queued and its keeper carries the state forward.  The shared stream's staging and rounds have to come behind those.
    for (AcrRun* r : live) {
        clc_ctx* c = r->ctx;
        if (c->stream == st) continue;
        const bool ok = (c->ev_group || hipEventCreateWithFlags(&c->ev_group, hipEventDisableTiming) == hipSuccess) &&
                        hipEventRecord(c->ev_group, c->stream) == hipSuccess && hipStreamWaitEvent(st, c->ev_group, 0) == hipSuccess;
        if (!ok) (void)hipStreamSynchronize(c->stream);
    }
    // the inputs of all solves: one staging launch per kMaxBatch of them
    for (size_t k = 0; k < live.size(); k += kMaxBatch) {
        const int n = (int)std::min<size_t>(kMaxBatch, live.size() - k);
        const double* src[kMaxBatch]; double* dst[kMaxBatch]; size_t cnt[kMaxBatch];
        for (int i = 0; i < n; ++i) { src[i] = live[k + i]->stage_src; dst[i] = live[k + i]->stage_dst; cnt[i] = live[k + i]->stage_n; }
        const hipError_t e = launch_acr_stage_chains(src, dst, cnt, n, st);
        if (e != hipSuccess) { fail_all(fail(ctx0, CLC_ERR_HIP, "acransac: shared staging launch", e)); return; }
    }
    // a round's launches carry the solves that are still in their rounds (what the host knows when it enqueues: a solve that finishes in
    // the round being waited for rides along once more and finds nothing to replay), kMaxBatch per launch
    auto enqueue = [&](const int bound) -> bool {
        AcrChains pack;
        int n = 0;
        auto flush = [&]() -> bool {
            if (n == 0) return true;
            const hipError_t e = kind != 1 ? launch_acr_round_p3p_chains(pack, n, launches & 1, bound, st)
                                           : launch_acr_round_5pt_chains(pack, n, launches & 1, bound, st);
            n = 0;
            if (e != hipSuccess) { fail_all(fail(ctx0, CLC_ERR_HIP, "acransac: shared round launch", e)); return false; }
            return true;
        };
        for (AcrRun* r : live) {
            if (r->phase != AcrRun::ROUNDS) continue;
            r->chain(pack.c[n++]);
            if (n == kMaxBatch && !flush()) return false;
        }
        if (!flush()) return false;
        ++launches;
        return true;
    };
    int b0 = 1, b1 = 1;
    for (AcrRun* r : live) { b0 = std::max(b0, r->first_bound); b1 = std::max(b1, r->bound); }
    if (!enqueue(b0) || !enqueue(b1)) return;                        // the second: speculative, the round after the one being waited for
    for (;;) {
        bool any_live = false, rounds_waiting = false, any_more = false;
        int bnd = 1;
        for (AcrRun* r : live) {
            if (r->phase == AcrRun::DONE) continue;
            any_live = true;
            if (!(r->phase == AcrRun::ROUNDS && r->reported)) (void)r->poll();   // ROUNDS: one look at the word; REFINE: one look at the record
            if (r->phase != AcrRun::ROUNDS) continue;
            if (!r->reported) rounds_waiting = true;                  // (in its rounds and reported = it needs another round)
            else { any_more = true; bnd = std::max(bnd, r->bound); }
        }
        if (!any_live) break;
        if (any_more && !rounds_waiting) {
            // every solve still in its rounds has reported the round waited for: the next shared launch (refinements of finished solves
            // may still be out on their own streams; they do not hold the rounds up)
            if (!enqueue(bnd)) return;
            for (AcrRun* r : live) if (r->phase == AcrRun::ROUNDS) r->advance();
        }
    }
    // The launch enqueued ahead of the last round is still in the shared stream (it finds nothing to replay, but its keepers carry every
    // chain's state forward inside that chain's workspace): whatever the other contexts enqueue next on THEIR streams -- the staging of
    // their next solve, a refinement -- has to come behind it.  (Work on the shared stream is ordered by the stream.)
    bool ordered = false;
    if (ctx0->ev_group || hipEventCreateWithFlags(&ctx0->ev_group, hipEventDisableTiming) == hipSuccess) {
        ordered = hipEventRecord(ctx0->ev_group, st) == hipSuccess;
        for (AcrRun* r : live)
            if (ordered && r->ctx->stream != st) ordered = hipStreamWaitEvent(r->ctx->stream, ctx0->ev_group, 0) == hipSuccess;
    }
    if (!ordered) (void)hipStreamSynchronize(st);
    // (runs of the batch that were not part of the group -- early outs -- are DONE already)
}

int acr_impl(clc_ctx* ctx, int kind, const double* h_a, const double* h_b, int N, const double* h_K1, const double* h_K2, int img_w,
             int img_h, int max_iteration, uint64_t seed, double precision, double refine_huber, double* h_model, uint8_t* h_mask,
             int32_t* h_inliers, int* n_inliers, double* error_max, double* min_nfa, int* iterations, int* rounds, double* h_cov,
             double* rmse)
{
    AcrRun run;
    run.ctx = ctx; run.kind = kind; run.h_a = h_a; run.h_b = h_b; run.N = N; run.h_K1 = h_K1; run.h_K2 = h_K2; run.img_w = img_w; run.img_h = img_h;
    run.max_iteration = max_iteration; run.seed = seed; run.precision = precision; run.refine_huber = refine_huber;
    run.h_model = h_model; run.h_mask = h_mask; run.h_inliers = h_inliers; run.n_inliers = n_inliers; run.error_max = error_max;
    run.min_nfa = min_nfa; run.iterations = iterations; run.rounds = rounds; run.h_cov = h_cov; run.rmse = rmse;
    int rc = run.begin();
    if (rc != CLC_OK) return rc;
    while (run.phase != AcrRun::DONE) {
        rc = run.poll();
        if (rc != CLC_OK) return rc;
    }
    run.finish();
    return run.status;
}

} // namespace

extern "C" {

int clc_pnp_acransac(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, int max_iteration, uint64_t seed,
                     double precision, double* h_Rt, uint8_t* h_inlier_mask, int32_t* h_inliers, int* n_inliers, double* error_max,
                     double* min_nfa, int* iterations)
{
    if (h_Rt) memset(h_Rt, 0, sizeof(double) * 12);
    return acr_impl(ctx, 0, h_X, h_x, N, h_K, nullptr, 0, 0, max_iteration, seed, precision, -1.0, h_Rt, h_inlier_mask, h_inliers, n_inliers,
                    error_max, min_nfa, iterations, nullptr, nullptr, nullptr);
}

int clc_pnp_localize_ac(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, int max_iteration, uint64_t seed,
                        double precision, double huber_a, double* h_Rt, double* h_cov, uint8_t* h_inlier_mask, int32_t* h_inliers,
                        int* n_inliers, double* error_max, double* rmse)
{
    if (h_Rt) memset(h_Rt, 0, sizeof(double) * 12);
    if (h_cov) memset(h_cov, 0, sizeof(double) * 36);
    if (rmse) *rmse = 0.0;
    return acr_impl(ctx, 0, h_X, h_x, N, h_K, nullptr, 0, 0, max_iteration, seed, precision, huber_a > 0.0 ? huber_a : 16.0, h_Rt,
                    h_inlier_mask, h_inliers, n_inliers, error_max, nullptr, nullptr, nullptr, h_cov, rmse);
}

int clc_pnp_localize_ac_batch(clc_ctx* const* ctxs, clc_pose_job* jobs, int n_jobs)
{
    if (n_jobs < 0 || (n_jobs > 0 && (!ctxs || !jobs))) return CLC_ERR_BAD_ARG;
    if (n_jobs == 0) return CLC_OK;
    for (int i = 0; i < n_jobs; ++i) {
        if (!ctxs[i]) return CLC_ERR_BAD_ARG;
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(ctxs[i], CLC_ERR_BAD_ARG, "pnp_localize_ac_batch: every job needs a context of its own");
        if (ctxs[i]->device != ctxs[0]->device) return fail(ctxs[i], CLC_ERR_BAD_ARG, "pnp_localize_ac_batch: the contexts must live on one device");
    }
    std::vector<AcrRun> runs((size_t)n_jobs);
    int worst = CLC_OK, live = 0;
    const bool lockstep = acr_lockstep(0, n_jobs);
    for (int i = 0; i < n_jobs; ++i) {
        clc_pose_job& jb = jobs[i];
        AcrRun& r = runs[(size_t)i];
        r.grouped = lockstep; r.group_stream = ctxs[0]->stream;
        if (jb.Rt) memset(jb.Rt, 0, sizeof(double) * 12);
        if (jb.cov) memset(jb.cov, 0, sizeof(double) * 36);
        jb.n_inliers = 0; jb.error_max = 0.0; jb.rmse = 0.0; jb.iterations = 0;
        r.ctx = ctxs[i]; r.kind = 0; r.h_a = jb.X; r.h_b = jb.x; r.N = jb.n; r.h_K1 = jb.K;
        r.max_iteration = jb.max_iteration; r.seed = jb.seed; r.precision = jb.precision;
        r.refine_huber = jb.refine ? (jb.huber_a > 0.0 ? jb.huber_a : 16.0) : -1.0;
        r.h_model = jb.Rt; r.h_mask = jb.inlier_mask; r.h_inliers = jb.inliers; r.n_inliers = &jb.n_inliers; r.error_max = &jb.error_max;
        r.iterations = &jb.iterations; r.h_cov = jb.cov; r.rmse = &jb.rmse;
        jb.status = r.begin();                        // stages this job's inputs (and, on its own, puts its first two rounds into its context's stream)
        if (r.phase != AcrRun::DONE) ++live;
    }
    (void)live;
    if (lockstep) drive_group(runs); else drive_runs(runs);
    for (int i = 0; i < n_jobs; ++i) {
        AcrRun& r = runs[(size_t)i];
        r.finish();
        jobs[i].status = r.status;
        if (r.status != CLC_OK && worst == CLC_OK) worst = r.status;
    }
    return worst;
}

int clc_essential_acransac(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1, const double* h_K2,
                           int img_w, int img_h, int max_iteration, uint64_t seed, double precision, double* h_E, double* h_F,
                           uint8_t* h_inlier_mask, int32_t* h_inliers, int* n_inliers, double* error_max, double* min_nfa, int* iterations)
{
    double EF[18] = {};
    const int rc = acr_impl(ctx, 1, h_x1, h_x2, N, h_K1, h_K2, img_w, img_h, max_iteration, seed, precision, -1.0, EF, h_inlier_mask,
                            h_inliers, n_inliers, error_max, min_nfa, iterations, nullptr, nullptr, nullptr);
    if (h_E) memcpy(h_E, EF, sizeof(double) * 9);
    if (h_F) memcpy(h_F, EF + 9, sizeof(double) * 9);
    return rc;
}

// RobustMatcher's model letter -> the solve kind (colocParams::model, RobustMatcher.hpp:399-405)
static int two_view_kind(const int model) { return model == CLC_MODEL_ESSENTIAL ? 1 : (model == CLC_MODEL_FUNDAMENTAL ? 2 : (model == CLC_MODEL_HOMOGRAPHY ? 3 : -1)); }

int clc_two_view_acransac(clc_ctx* ctx, int model, const double* h_x1, const double* h_x2, int N, const double* h_K1, const double* h_K2,
                          int img_w, int img_h, int max_iteration, uint64_t seed, double precision, double* h_M, double* h_F,
                          uint8_t* h_inlier_mask, int32_t* h_inliers, int* n_inliers, double* error_max, double* min_nfa, int* iterations)
{
    const int kind = two_view_kind(model);
    if (kind < 0) return fail(ctx, CLC_ERR_BAD_ARG, "two_view_acransac: model must be 'E', 'F' or 'H'");
    if (kind == 1)
        return clc_essential_acransac(ctx, h_x1, h_x2, N, h_K1, h_K2, img_w, img_h, max_iteration, seed, precision, h_M, h_F, h_inlier_mask,
                                      h_inliers, n_inliers, error_max, min_nfa, iterations);
    double M9[9] = {};
    const int rc = acr_impl(ctx, kind, h_x1, h_x2, N, nullptr, nullptr, img_w, img_h, max_iteration, seed, precision, -1.0, M9, h_inlier_mask,
                            h_inliers, n_inliers, error_max, min_nfa, iterations, nullptr, nullptr, nullptr);
    if (h_M) memcpy(h_M, M9, sizeof M9);
    if (h_F) { if (kind == 2) memcpy(h_F, M9, sizeof M9); else memset(h_F, 0, sizeof M9); }
    return rc;
}

int clc_two_view_minimal(clc_ctx* ctx, int model, const double* h_x1, const double* h_x2, int N, int img_w, int img_h, const int32_t* h_samples,
                         int S, double* h_models)
{
    const int kind = two_view_kind(model);
    if (!ctx || kind < 2 || N <= 0 || S < 0 || !h_x1 || !h_x2 || img_w <= 0 || img_h <= 0 || (S > 0 && (!h_samples || !h_models)))
        return fail(ctx, CLC_ERR_BAD_ARG, "two_view_minimal: bad argument (models 'F' and 'H')");
    if (S == 0) return CLC_OK;
    const int m = kind == 2 ? 7 : 4, M = kind == 2 ? 3 : 1;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t pts = (size_t)4 * N, smp = dbl(sizeof(int32_t) * (size_t)S * m), out = (size_t)S * M * 9;
    int rc = ensure_pnp(ctx, pts + smp + out + 8);
    if (rc != CLC_OK) return rc;
    std::vector<double> q(pts);
    const tv::Normalizer t = tv::normalizer(img_w, img_h);
    for (int i = 0; i < N; ++i) {
        q[2 * i] = h_x1[2 * i] * t.d + t.tx; q[2 * i + 1] = h_x1[2 * i + 1] * t.d + t.ty;
        q[(size_t)2 * N + 2 * i] = h_x2[2 * i] * t.d + t.tx; q[(size_t)2 * N + 2 * i + 1] = h_x2[2 * i + 1] * t.d + t.ty;
    }
    double* d = ctx->d_pnp;
    int32_t* d_smp = (int32_t*)(d + pts);
    double* d_out = d + pts + smp;
    CLC_HIP(ctx, hipMemcpyAsync(d, q.data(), sizeof(double) * pts, hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, hipMemcpyAsync(d_smp, h_samples, sizeof(int32_t) * (size_t)S * m, hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, launch_twoview_minimal(kind, d, d + (size_t)2 * N, N, d_smp, S, d_out, ctx->stream));
    CLC_HIP(ctx, hipMemcpyAsync(h_models, d_out, sizeof(double) * out, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

} // extern "C"

namespace {

void two_view_begin(AcrRun& r, clc_ctx* ctx, clc_two_view_job& jb, double* EF, const bool lockstep, hipStream_t group_stream, const int kind)
{
    r.grouped = lockstep; r.group_stream = group_stream;
    if (jb.E) memset(jb.E, 0, sizeof(double) * 9);
    if (jb.F) memset(jb.F, 0, sizeof(double) * 9);
    jb.n_inliers = 0; jb.iterations = 0; jb.error_max = 0.0; jb.min_nfa = INFINITY;
    r.ctx = ctx; r.kind = kind; r.h_a = jb.x1; r.h_b = jb.x2; r.N = jb.n; r.h_K1 = jb.K1; r.h_K2 = jb.K2; r.img_w = jb.img_w; r.img_h = jb.img_h;
    r.max_iteration = jb.max_iteration; r.seed = jb.seed; r.precision = jb.precision; r.refine_huber = -1.0;
    r.h_model = EF; r.h_mask = jb.inlier_mask; r.h_inliers = jb.inliers; r.n_inliers = &jb.n_inliers; r.error_max = &jb.error_max;
    r.min_nfa = &jb.min_nfa; r.iterations = &jb.iterations;
    jb.status = r.begin();
}


} // namespace

namespace clc {

int check_batch_contexts(clc_ctx* const* ctxs, int n_jobs, const char* what)
{
    for (int i = 0; i < n_jobs; ++i) {
        if (!ctxs[i]) return CLC_ERR_BAD_ARG;
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(ctxs[i], CLC_ERR_BAD_ARG, what);
        if (ctxs[i]->device != ctxs[0]->device) return fail(ctxs[i], CLC_ERR_BAD_ARG, "batch: the contexts must live on one device");
    }
    return CLC_OK;
}


// kind 1: E -> job.E, F -> job.F; kinds 2 / 3: the model matrix -> job.E (RelativePose_Info::essential_matrix receives it, whatever the
// model: RobustMatcher.hpp:141, :207), and job.F = F for kind 2, zeros for kind 3
int acr_two_view_batch(clc_ctx* const* ctxs, clc_two_view_job* const* jobs, int n_jobs, int kind)
{
    std::vector<AcrRun> runs((size_t)n_jobs);
    std::vector<double> EF((size_t)18 * n_jobs, 0.0);
    const bool lockstep = acr_lockstep(kind, n_jobs);
    for (int i = 0; i < n_jobs; ++i) two_view_begin(runs[(size_t)i], ctxs[i], *jobs[i], &EF[(size_t)18 * i], lockstep, ctxs[0]->stream, kind);
    if (lockstep) drive_group(runs); else drive_runs(runs);
    int worst = CLC_OK;
    for (int i = 0; i < n_jobs; ++i) {
        AcrRun& r = runs[(size_t)i];
        r.finish();
        jobs[i]->status = r.status;
        if (jobs[i]->E) memcpy(jobs[i]->E, &EF[(size_t)18 * i], sizeof(double) * 9);
        if (jobs[i]->F) memcpy(jobs[i]->F, &EF[(size_t)18 * i + (kind == 1 ? 9 : (kind == 2 ? 0 : 9))], sizeof(double) * 9);
        if (r.status != CLC_OK && worst == CLC_OK) worst = r.status;
    }
    return worst;
}

} // namespace clc

extern "C" {

int clc_essential_acransac_batch(clc_ctx* const* ctxs, clc_two_view_job* jobs, int n_jobs)
{
    if (n_jobs < 0 || (n_jobs > 0 && (!ctxs || !jobs))) return CLC_ERR_BAD_ARG;
    if (n_jobs == 0) return CLC_OK;
    const int rc0 = check_batch_contexts(ctxs, n_jobs, "essential_acransac_batch: every job needs a context of its own");
    if (rc0 != CLC_OK) return rc0;
    std::vector<clc_two_view_job*> ptr((size_t)n_jobs);
    for (int i = 0; i < n_jobs; ++i) ptr[(size_t)i] = &jobs[i];
    return acr_two_view_batch(ctxs, ptr.data(), n_jobs, 1);
}

int clc_two_view_acransac_batch(clc_ctx* const* ctxs, int model, clc_two_view_job* jobs, int n_jobs)
{
    const int kind = two_view_kind(model);
    if (kind < 0 || n_jobs < 0 || (n_jobs > 0 && (!ctxs || !jobs))) return CLC_ERR_BAD_ARG;
    if (n_jobs == 0) return CLC_OK;
    const int rc0 = check_batch_contexts(ctxs, n_jobs, "two_view_acransac_batch: every job needs a context of its own");
    if (rc0 != CLC_OK) return rc0;
    std::vector<clc_two_view_job*> ptr((size_t)n_jobs);
    for (int i = 0; i < n_jobs; ++i) ptr[(size_t)i] = &jobs[i];
    return acr_two_view_batch(ctxs, ptr.data(), n_jobs, kind);
}

} // extern "C"
