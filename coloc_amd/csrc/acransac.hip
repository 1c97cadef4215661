// acransac.hip -- a-contrario RANSAC (AC-RANSAC) on gfx950: the model selection OpenMVG runs for the reference's pose
// and two-view steps,
//     SfM_Localizer::Localize(P3P_KE_CVPR17, ..., {error_max = +inf, max_iteration = 256})   include/coloc/Localizer.hpp:82-93
//     ACRANSAC(ACKernelAdaptorEssential<FivePointSolver, SymmetricEpipolarDistanceError>, ...)  include/coloc/RobustMatcher.hpp:153-171
// restated from the published algorithm (Moisan, Moulon, Monasse, IPOL 2012; see oracle/clc_oracle_acr.c for the
// sequential form and what is unpinned).  Per model: residuals over ALL data, sorted with their indices, NFA(k) over
// the k smallest, the model's value = min_k NFA(k); the run keeps the model with the lowest value, and once a
// meaningful one (NFA < 0) exists the remaining 10 % reserve of iterations sample among its inliers.
//
// GPU shape.  The sequential loop evaluates <= 4 (10) models per iteration, one after the other, each with an O(n log n)
// sort on one CPU thread.  Here a ROUND evaluates a batch of B iterations at once:
//   solve   : the existing minimal solvers (p3p_kernel: one problem per four lanes; fivept_kernel: one per wave) on the
//             batch's samples -- samples are a pure function of (seed, iteration, index set), see clc_acr.h;
//   nfa     : one workgroup per model slot: residuals straight into LDS, bitonic sort of (residual bits, index) in LDS,
//             NFA(k) for every k in parallel, min-reduction -> {nfa, k, e_k} and the sorted index list;
//   select  : one workgroup replays the SEQUENTIAL semantics over the batch in iteration / solver order (strict '<'
//             improvements, the phase-switch rule), stops at the first iteration that changes the index set -- the
//             iterations after it were sampled speculatively from the old set and are discarded --, updates the device
//             state, draws the next batch's samples and mirrors the state into pinned host memory.
// The host only reads that mirror (one stream synchronisation per round) to learn whether another round is needed:
// typically one round to find the first meaningful model, one or two for the reserve.  Results are identical to the
// sequential oracle: same samples, bit-identical residuals (same operation order, no FMA contraction), a total order on
// (residual, index), and the same portable log10 in the NFA terms.
#include "clc_internal.h"
#include "clc_acr.h"

namespace clc {

// where the round that COMPLETES a run leaves the result (round 3: the finish work rides in that round's select launch, and the
// host returns as soon as the polled word says "done" instead of launching a finish kernel behind the round enqueued ahead)
struct AcrFinish {
    uint8_t* d_mask; AcrResult* d_res;                 // device copies (the refinement reads them)
    uint8_t* h_mask; int32_t* h_inliers; AcrResult* h_res;     // pinned host memory (nullable)
};
__device__ __forceinline__ void acr_finish_block(const AcrProblem& pb, const AcrState& s, const uint32_t* __restrict__ best_inliers,
                                                 const AcrFinish& fin, const int tid, const int T)
{
    const bool ok = s.min_nfa < 0.0 && s.n_inliers > 0;
    const int n_inl = ok ? s.n_inliers : 0;
    for (int i = tid; i < pb.n; i += T) { fin.d_mask[i] = 0; if (fin.h_mask) fin.h_mask[i] = 0; }
    __syncthreads();
    for (int i = tid; i < n_inl; i += T) {
        const uint32_t p = best_inliers[i];
        fin.d_mask[p] = 1;
        if (fin.h_mask) fin.h_mask[p] = 1;
        if (fin.h_inliers) fin.h_inliers[i] = (int32_t)p;
    }
    if (tid == 0) {
        AcrResult r;
        for (int e = 0; e < 18; ++e) r.model[e] = ok ? s.model[e] : 0.0;
        r.min_nfa = s.min_nfa;
        // unormalizeError: resection sqrt(e) / N1(0,0) -> pixels; essential: the squared pixel distance as it is
        r.error_max = !ok ? 0.0 : (pb.kind == 0 ? sqrt(s.error_max) / pb.norm : s.error_max);
        r.n_inliers = n_inl;
        r.valid = ok ? s.best_iter : -1;
        r.iterations = s.iter;
        r.rounds = s.rounds_eval;               // not the sequence number: the round enqueued ahead of the host's knowledge is empty
        *fin.d_res = r;
        if (fin.h_res) *fin.h_res = r;
    }
}

struct AcrHyp {            // per model slot, written by the nfa kernel
    double nfa;            // min_k NFA(k); +inf for an empty slot
    double e_k;            // the k-th smallest residual (kernel units)
    int32_t k;             // minimising k
    int32_t n_le;          // residuals <= max_threshold (upper-bound mode gate)
};

// ---- nfa: one workgroup per model slot ---------------------------------------------------------------------------
__device__ __forceinline__ double acr_err_resection(const double* __restrict__ P, const double* __restrict__ K, const double s,
                                                    const double Xw, const double Yw, const double Zw, const double uo, const double vo)
{
    // ACKernelAdaptorResection_Intrinsics::Errors: (pixel residual * 1 / focal).squaredNorm(); operation order of
    // oracle/clc_oracle_acr.c acr_errors
    const double xc = ((P[0] * Xw + P[1] * Yw) + P[2] * Zw) + P[3];
    const double yc = ((P[4] * Xw + P[5] * Yw) + P[6] * Zw) + P[7];
    const double zc = ((P[8] * Xw + P[9] * Yw) + P[10] * Zw) + P[11];
    const double u = (K[0] * xc + K[1] * yc) + K[2] * zc;
    const double v = (K[3] * xc + K[4] * yc) + K[5] * zc;
    const double w = (K[6] * xc + K[7] * yc) + K[8] * zc;
    const double du = (uo - u / w) * s;
    const double dv = (vo - v / w) * s;
    return du * du + dv * dv;
}

__device__ __forceinline__ double acr_err_epipolar(const double* __restrict__ f, const double u1, const double v1, const double u2, const double v2)
{
    const double a0 = (f[0] * u1 + f[1] * v1) + f[2];
    const double a1 = (f[3] * u1 + f[4] * v1) + f[5];
    const double a2 = (f[6] * u1 + f[7] * v1) + f[8];
    const double b0 = (f[0] * u2 + f[3] * v2) + f[6];
    const double b1 = (f[1] * u2 + f[4] * v2) + f[7];
    const double d = (u2 * a0 + v2 * a1) + a2;
    return (d * d) * (1.0 / (a0 * a0 + a1 * a1) + 1.0 / (b0 * b0 + b1 * b1)) / 4.0;
}

// (residual bits, index) pairs in lexicographic order: what std::sort does with pair<double, uint32_t>
struct AcrItem { uint64_t key; uint32_t idx; };
__device__ __forceinline__ bool acr_gt(const uint64_t ka, const uint32_t ia, const uint64_t kb, const uint32_t ib)
{
    return ka > kb || (ka == kb && ia > ib);
}

// Sorting the residuals of one model.  Each thread holds E consecutive elements in REGISTERS: compare-exchange distances
// below E stay inside the thread, distances below 64 E are lane exchanges inside the wave (ds_bpermute, no barrier), only
// distances that cross waves go through LDS (1024 elements on 256 threads: 19 in-register, 33 in-wave, 3 cross-wave steps
// instead of 55 LDS round trips with a workgroup barrier each).
// The network runs on ONE 64-bit word per element, compared with v_min_f64 / v_max_f64 (two instructions per exchange
// instead of a 64-bit compare, an index compare and three selects): the residual's bits with the low 13 mantissa bits
// replaced by the element index (13 bits for n <= 8192, 14 up to 16 384).  That orders by (top 51 bits of the residual, index); the exact (residual,
// index) order differs from it only where two residuals agree in their top 51 bits, so afterwards every element recomputes
// its exact residual, neighbours are compared exactly, and in the (rare) case of an inversion anywhere in the workgroup
// the exact-key network below re-sorts -- the result is always the exact lexicographic order.
__device__ __forceinline__ double acr_fmin(const double a, const double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double acr_fmax(const double a, const double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double acr_shfl_xor(const double v, const int mask)
{
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = __shfl_xor((uint32_t)u, mask), hi = __shfl_xor((uint32_t)(u >> 32), mask);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// lane ^ D exchange of one dword over DPP (vector-ALU moves, a few cycles) for the distances that have a DPP form; ds_bpermute
// (an LDS-crossbar round trip, ~100 cycles) otherwise (v_permlane16/32_swap for 16 and 32 measured no better).  The sort is a chain of 55 dependent steps per wave and 34 of its 45
// in-wave steps have distance 1, 2, 4 or 8 (ablation: the sort is 8 of the kernel's 14.8 us, and halving its vector instruction
// count changed nothing -- it is latency per step).  D is a template argument: a run-time switch in every step cost more than
// the DPP moves saved.
template <int D>
__device__ __forceinline__ uint32_t acr_lane_xor(const uint32_t v)
{
    if (D == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    if (D == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    if (D == 4) {
        int r = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);                      // row_shl:4 into lanes 0-3, 8-11 of a row
        r = __builtin_amdgcn_update_dpp(r, (int)v, 0x114, 0xF, 0xA, false);                          // row_shr:4 into lanes 4-7, 12-15
        return (uint32_t)r;
    }
    if (D == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);      // row_ror:8
    return __shfl_xor(v, D);
}
// one in-wave compare-exchange step of the composite-key network: partner thread tid ^ DT (DT < 64), same slot
template <int E, int DT>
__device__ __forceinline__ void acr_step_wave(double (&c)[E], const int k, const int tid)
{
    const bool lower = (tid & DT) == 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint64_t u = (uint64_t)__double_as_longlong(c[e]);
        const uint32_t lo32 = acr_lane_xor<DT>((uint32_t)u), hi32 = acr_lane_xor<DT>((uint32_t)(u >> 32));
        const double o = __longlong_as_double((long long)(((uint64_t)hi32 << 32) | lo32));
        const bool keep_min = lower == (((tid * E + e) & k) == 0);
        const double lo = acr_fmin(c[e], o), hi = acr_fmax(c[e], o);
        c[e] = keep_min ? lo : hi;
    }
}

template <int E, bool EXACT>
__device__ __forceinline__ void acr_bitonic(uint64_t (&key)[E], uint32_t (&idx)[E], double (&c)[E], const int P, const int tid, const int T,
                                            uint64_t* lkey, uint32_t* lidx)
{
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < E) {
                // both elements in this thread: element e pairs with e | j
#pragma unroll
                for (int jj = 1; jj < E; jj <<= 1) {
                    if (jj != j) continue;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if (e & jj) continue;
                        const bool up = ((tid * E + e) & k) == 0;
                        if (EXACT) {
                            const bool gt = acr_gt(key[e], idx[e], key[e | jj], idx[e | jj]);
                            if (gt == up) {
                                const uint64_t tk = key[e]; key[e] = key[e | jj]; key[e | jj] = tk;
                                const uint32_t ti = idx[e]; idx[e] = idx[e | jj]; idx[e | jj] = ti;
                            }
                        } else {
                            const double lo = acr_fmin(c[e], c[e | jj]), hi = acr_fmax(c[e], c[e | jj]);
                            c[e] = up ? lo : hi;
                            c[e | jj] = up ? hi : lo;
                        }
                    }
                }
            } else if (!EXACT && j / E < 64) {
                // the in-wave tail of a merge is always the same run of distances 32, 16, .. 1 from wherever it starts: straight-line
                // code with compile-time distances, entered at the right place (a not-taken skip costs a cycle; a switch per step
                // costs more than the DPP moves save)
                const int d0 = j / E;
                if (d0 >= 32) acr_step_wave<E, 32>(c, k, tid);
                if (d0 >= 16) acr_step_wave<E, 16>(c, k, tid);
                if (d0 >= 8) acr_step_wave<E, 8>(c, k, tid);
                if (d0 >= 4) acr_step_wave<E, 4>(c, k, tid);
                if (d0 >= 2) acr_step_wave<E, 2>(c, k, tid);
                acr_step_wave<E, 1>(c, k, tid);
                j = E;                                                   // the loop continues with the in-register steps (j < E), if any
            } else {
                const int dt = j / E;                                    // partner thread = tid ^ dt, same e
                const bool lower = (tid & dt) == 0;
                uint32_t oidx[E];                                         // EXACT, cross-wave: the partner's indices (second pass through the buffer)
                if (dt >= 64) {
                    if (EXACT) {
                        // the rare exact re-sort stages the indices first, then the keys, through the SAME words: the staging area holds
                        // one 8-byte word per element (128 KB at 16 384 elements; keys + indices side by side would not fit)
                        __syncthreads();
#pragma unroll
                        for (int e = 0; e < E; ++e) lkey[e * T + tid] = (uint64_t)idx[e];
                        __syncthreads();
#pragma unroll
                        for (int e = 0; e < E; ++e) oidx[e] = (uint32_t)lkey[e * T + (tid ^ dt)];
                    }
                    __syncthreads();                                      // the previous exchange's reads are done
#pragma unroll
                    for (int e = 0; e < E; ++e) lkey[e * T + tid] = EXACT ? key[e] : (uint64_t)__double_as_longlong(c[e]);
                    __syncthreads();
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const bool keep_min = lower == (((tid * E + e) & k) == 0);
                    if (EXACT) {
                        uint64_t ok;
                        uint32_t oi;
                        if (dt >= 64) { ok = lkey[e * T + (tid ^ dt)]; oi = oidx[e]; }
                        else {
                            const uint32_t lo = __shfl_xor((uint32_t)key[e], dt), hi = __shfl_xor((uint32_t)(key[e] >> 32), dt);
                            ok = ((uint64_t)hi << 32) | lo;
                            oi = __shfl_xor(idx[e], dt);
                        }
                        const bool gt = acr_gt(key[e], idx[e], ok, oi);   // mine > other
                        if (gt == keep_min) { key[e] = ok; idx[e] = oi; }
                    } else {
                        const double o = dt >= 64 ? __longlong_as_double((long long)lkey[e * T + (tid ^ dt)]) : acr_shfl_xor(c[e], dt);
                        const double lo = acr_fmin(c[e], o), hi = acr_fmax(c[e], o);
                        c[e] = keep_min ? lo : hi;
                    }
                }
            }
        }
    }
}

template <int E>
__global__ __launch_bounds__(1024) void acr_nfa_kernel(const AcrProblem pb, const int P /* = blockDim.x * E, power of two >= n */,
                                                       const double* __restrict__ models, AcrHyp* __restrict__ hyp,
                                                       uint32_t* __restrict__ sorted_idx, const AcrState* __restrict__ state)
{
    extern __shared__ unsigned char acr_lds[];
    const int slot = blockIdx.x, tid = threadIdx.x, T = blockDim.x, n = pb.n;
    if (slot >= state->cur_batch * pb.max_models) return;          // the grid covers the largest batch; this round is smaller
    uint64_t* lkey = reinterpret_cast<uint64_t*>(acr_lds);               // [e][tid] staging of the cross-wave exchanges (one word per element)
    uint32_t* lidx = nullptr;
    __shared__ double s_nfa[1024 / 64], s_ek[1024 / 64];
    __shared__ int s_k[1024 / 64], s_cnt[1024 / 64];
    __shared__ uint64_t s_edge_key[1024 / 64];
    __shared__ uint32_t s_edge_idx[1024 / 64];
    const double* model = models + (size_t)slot * pb.model_doubles;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    // an empty slot (the solver marks it with NaNs) never improves anything
    if (model[0] != model[0]) {
        if (tid == 0) { hyp[slot].nfa = inf; hyp[slot].e_k = 0.0; hyp[slot].k = 0; hyp[slot].n_le = 0; }
        return;
    }
    auto residual = [&](const int i) -> double {
        return pb.kind == 0
            ? acr_err_resection(model, pb.K1, pb.norm, pb.a[3 * i], pb.a[3 * i + 1], pb.a[3 * i + 2], pb.b[2 * i], pb.b[2 * i + 1])
            : acr_err_epipolar(model, pb.a[2 * i], pb.a[2 * i + 1], pb.b[2 * i], pb.b[2 * i + 1]);
    };
    uint64_t key[E];
    uint32_t idx[E];
    double c[E];
    int cnt = 0;
    const uint64_t kMaxFinite = 0x7fefffffffffffffull;
    // the composite sort key keeps the residual's top bits and carries the element index in the low 13 (n <= 8192) or 14 (n <= 16384) ones
    constexpr uint64_t kIdxMask = E > 8 ? 0x3FFFull : 0x1FFFull;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e;
        uint64_t bits = kMaxFinite;                                       // padding (and inf / NaN residuals) sort behind every finite one
        if (i < n) {
            const double r = residual(i);
            cnt += r <= pb.max_threshold ? 1 : 0;
            const uint64_t rb = (uint64_t)__double_as_longlong(r);
            bits = rb < kMaxFinite ? rb : kMaxFinite;
        }
        c[e] = __longlong_as_double((long long)((bits & ~kIdxMask) | (uint64_t)i));
    }
    acr_bitonic<E, false>(key, idx, c, P, tid, T, lkey, lidx);
    // exact keys of the elements as they stand now, then the neighbour check
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)((uint64_t)__double_as_longlong(c[e]) & kIdxMask);
        if ((int)i < n) { idx[e] = i; key[e] = (uint64_t)__double_as_longlong(residual((int)i)); }
        else { idx[e] = 0xFFFFFFFFu; key[e] = 0x7ff0000000000000ull; }   // padding: +inf, index above every real one
    }
    bool bad = false;
#pragma unroll
    for (int e = 0; e + 1 < E; ++e) bad = bad || acr_gt(key[e], idx[e], key[e + 1], idx[e + 1]);
    {
        // my last element against the next thread's first: inside the wave by lane shift, across waves through LDS
        const uint32_t nlo = __shfl_down((uint32_t)key[0], 1), nhi = __shfl_down((uint32_t)(key[0] >> 32), 1);
        const uint32_t nidx = __shfl_down(idx[0], 1);
        if ((tid & 63) == 0) { s_edge_key[tid >> 6] = key[0]; s_edge_idx[tid >> 6] = idx[0]; }
        __syncthreads();
        uint64_t nk = ((uint64_t)nhi << 32) | nlo;
        uint32_t ni = nidx;
        if ((tid & 63) == 63) {
            if (tid + 1 < T) { nk = s_edge_key[(tid >> 6) + 1]; ni = s_edge_idx[(tid >> 6) + 1]; }
            else { nk = ~0ull; ni = ~0u; }
        }
        bad = bad || acr_gt(key[E - 1], idx[E - 1], nk, ni);
    }
    if (__syncthreads_or(bad ? 1 : 0)) acr_bitonic<E, true>(key, idx, c, P, tid, T, lkey, lidx);
    // NFA(k) of this thread's own positions k = tid E + e + 1, for m + 1 <= k <= n and e_(k) <= max_threshold; strict '<'
    // keeps the first k
    double best = inf, bek = 0.0;
    int bk = pb.m;
    uint32_t* out_idx = sorted_idx + (size_t)slot * n;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int kk = tid * E + e + 1;
        if (kk <= n) out_idx[kk - 1] = idx[e];
        if (kk > pb.m && kk <= n) {
            const double r = __longlong_as_double((long long)key[e]);
            if (r <= pb.max_threshold) {
                const double v = clc_acr_nfa(pb.loge0, pb.logalpha0, pb.mult, r, kk, pb.m, pb.logc_n[kk], pb.logc_k[kk]);
                if (v < best) { best = v; bk = kk; bek = r; }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off), oe = __shfl_xor(bek, off);
        const int ok = __shfl_xor(bk, off);
        if (ov < best || (ov == best && ok < bk)) { best = ov; bk = ok; bek = oe; }
        cnt += __shfl_xor(cnt, off);
    }
    if ((tid & 63) == 0) { s_nfa[tid >> 6] = best; s_k[tid >> 6] = bk; s_ek[tid >> 6] = bek; s_cnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        int total = 0;
        best = inf; bk = pb.m; bek = 0.0;
        for (int w = 0; w < (T + 63) / 64; ++w) {
            if (s_nfa[w] < best || (s_nfa[w] == best && s_k[w] < bk)) { best = s_nfa[w]; bk = s_k[w]; bek = s_ek[w]; }
            total += s_cnt[w];
        }
        hyp[slot].nfa = best;
        hyp[slot].k = bk;
        hyp[slot].e_k = bek;
        hyp[slot].n_le = total;
    }
}

// ---- select: the sequential semantics over one batch -------------------------------------------------------------
// block-wide reductions of one int / the inclusive min-scan of one double over 256 threads: wave shuffles + one LDS hop
__device__ __forceinline__ int acr_block_reduce(int v, const bool take_min, int* s_red, const int tid)
{
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = take_min ? (o < v ? o : v) : (o > v ? o : v); }
    __syncthreads();                                               // s_red may still be read from the previous call
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    int r = s_red[0];
    for (int w = 1; w < 4; ++w) r = take_min ? (s_red[w] < r ? s_red[w] : r) : (s_red[w] > r ? s_red[w] : r);
    return r;
}
__device__ __forceinline__ double acr_block_scan_min(double v, double* s_part, const int tid)
{
    const int lane = tid & 63;
    for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(v, off); if (lane >= off && o < v) v = o; }
    __syncthreads();
    if (lane == 63) s_part[tid >> 6] = v;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); ++w) v = s_part[w] < v ? s_part[w] : v;
    return v;                                                       // inclusive prefix minimum over threads 0..tid
}

__global__ __launch_bounds__(256) void acr_select_kernel(const AcrProblem pb, const double* __restrict__ models,
                                                         const AcrHyp* __restrict__ hyp, const uint32_t* __restrict__ sorted_idx,
                                                         AcrState* __restrict__ state, uint32_t* __restrict__ best_inliers,
                                                         uint32_t* __restrict__ index_set, int32_t* __restrict__ samples,
                                                         unsigned long long* __restrict__ h_word, const AcrFinish fin)
{
    // The loop being replayed is sequential (strict '<' improvements in iteration / solver order, the first iteration
    // that switches the index set ends the batch), but everything in it is a prefix operation over the <= 1280 model
    // slots: the running minimum is a prefix min, "improved" compares a slot with the prefix before it, the batch ends
    // at the FIRST iteration whose condition holds.  One workgroup; the kernel is a chain of dependent memory round
    // trips, so it is written to have as few of them as possible: (state + slots) -> (winner's index list) -> stores.
    constexpr int kMaxSlots = kAcrMaxBatch * 10;
    constexpr int T = 256;
    constexpr int kPer = (kMaxSlots + T - 1) / T;
    __shared__ AcrState s;
    __shared__ double s_pre[kMaxSlots];                            // prefix min INCLUDING the slot
    __shared__ unsigned char s_imp[kMaxSlots];
    __shared__ double s_part[4];
    __shared__ int s_red[4];
    __shared__ int s_best_h, s_copy_index;
    const int tid = threadIdx.x;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    // round trip 1: the state and this thread's slots (loaded for the largest batch; masked below)
    const int cur_batch = state->cur_batch, ac_mode0 = state->ac_mode, iter0 = state->iter, n_iter0 = state->n_iter, reserve0 = state->reserve;
    const double min0 = state->min_nfa;
    const int B = cur_batch, total = B * pb.max_models;
    const int C = (total + T - 1) / T, h0 = tid * C;
    double val[kPer];
    int nle[kPer];
#pragma unroll
    for (int c = 0; c < kPer; ++c) {
        const int h = h0 + c;
        const bool in = c < C && h < total;
        val[c] = in ? hyp[h].nfa : inf;
        nle[c] = in ? hyp[h].n_le : 0;
    }
    if (tid == 0) s = *state;
    // 1. upper-bound mode gate: slots before the first model with more than 2.5 m residuals under the bound are ignored
    int first_gate = kMaxSlots;
#pragma unroll
    for (int c = kPer - 1; c >= 0; --c) if ((double)nle[c] > 2.5 * (double)pb.m && h0 + c < total) first_gate = h0 + c;
    const int first_on = ac_mode0 ? 0 : acr_block_reduce(first_gate, true, s_red, tid);
    // 2. prefix minimum of the slot values, seeded with the minimum of the previous rounds
    double lm = inf;
#pragma unroll
    for (int c = 0; c < kPer; ++c) {
        if (h0 + c < first_on) val[c] = inf;
        lm = val[c] < lm ? val[c] : lm;
    }
    const double incl = acr_block_scan_min(lm, s_part, tid);
    double run = __shfl_up(incl, 1);
    if ((tid & 63) == 0) run = tid == 0 ? inf : inf;               // previous wave's total comes through s_part below
    if ((tid & 63) == 0 && tid > 0) { run = inf; for (int w = 0; w < (tid >> 6); ++w) run = s_part[w] < run ? s_part[w] : run; }
    run = min0 < run ? min0 : run;
#pragma unroll
    for (int c = 0; c < kPer; ++c) {
        const int h = h0 + c;
        if (c < C && h < total) {
            s_imp[h] = val[c] < run ? 1 : 0;                       // strict: an equal value does not replace the earlier model
            run = val[c] < run ? val[c] : run;
            s_pre[h] = run;
        }
    }
    __syncthreads();
    // 3. the first iteration that ends the batch
    int ev = B;
    for (int it = tid; it < B; it += T) {
        bool better = false;
        for (int k = 0; k < pb.max_models; ++k) better = better || s_imp[it * pb.max_models + k];
        const double min_after = s_pre[it * pb.max_models + pb.max_models - 1];
        const int cur = iter0 + it;
        if (((better && min_after < 0.0) || (cur + 1 == n_iter0 && reserve0)) && it < ev) ev = it;
    }
    const int event_it = acr_block_reduce(ev, true, s_red, tid), consumed = event_it < B ? event_it + 1 : B;
    // 4. the last improvement among the consumed slots is the model the sequential loop ends up with
    int last_imp = -1;
#pragma unroll
    for (int c = 0; c < kPer; ++c) { const int h = h0 + c; if (c < C && h < consumed * pb.max_models && s_imp[h]) last_imp = h; }
    const int best_red = acr_block_reduce(last_imp, false, s_red, tid);
    if (tid == 0) {
        const int best_h = best_red;
        int copy_index = 0;
        if (!s.ac_mode && first_on < consumed * pb.max_models) s.ac_mode = 1;
        if (best_h >= 0) {
            const AcrHyp hy = hyp[best_h];
            s.min_nfa = hy.nfa;
            s.n_inliers = hy.k;
            s.error_max = hy.e_k;
            s.best_iter = s.iter + best_h / pb.max_models;
            for (int e = 0; e < pb.model_doubles; ++e) s.model[e] = models[(size_t)best_h * pb.model_doubles + e];
        }
        if (event_it < B) {
            const int cur = s.iter + event_it;
            if (s.n_inliers == 0) { s.n_iter++; s.reserve--; }
            else {
                copy_index = 1;
                s.n_index = s.n_inliers;
                s.index_all = 0;
                if (s.reserve) { s.n_iter = cur + 1 + s.reserve; s.reserve = 0; }
            }
        }
        s.iter += consumed;
        if (B > 0) { s.rounds += 1; s.rounds_eval += 1; }      // an empty round (enqueued ahead, after the end) is not a round
        s.last_batch = consumed;
        // next round: while nothing has happened look further ahead per round; after an event the whole reserve goes in one
        if (event_it < B || !s.index_all) s.grow = kAcrMaxBatch;
        else s.grow = s.grow * 2 > kAcrMaxBatch ? kAcrMaxBatch : s.grow * 2;
        const int remaining = s.n_iter - s.iter;
        s.cur_batch = remaining < s.grow ? (remaining > 0 ? remaining : 0) : s.grow;
        s_best_h = best_h;
        s_copy_index = copy_index;
    }
    __syncthreads();
    // round trip 2: the winner's sorted index list feeds best_inliers, (on a switch) the index set, and the next samples
    const int best_h = s_best_h, n_inl = s.n_inliers, copy_index = s_copy_index;
    const uint32_t* win = best_h >= 0 ? sorted_idx + (size_t)best_h * pb.n : nullptr;
    if (win)
        for (int i = tid; i < n_inl; i += T) {
            const uint32_t v = win[i];
            best_inliers[i] = v;
            if (copy_index) index_set[i] = v;
        }
    else if (copy_index)                                // vec_index = vec_inliers of a model found in an earlier round
        for (int i = tid; i < n_inl; i += T) index_set[i] = best_inliers[i];
    // the samples of the next round, from the index set as it now stands: positions map through the list just chosen
    // (read from where it came from, not from the copy being written)
    {
        const uint32_t* src = s.index_all ? nullptr : (copy_index ? (win ? win : best_inliers) : index_set);
        const int remaining = s.n_iter - s.iter;
        const int nb = remaining < kAcrMaxBatch ? remaining : kAcrMaxBatch;
        for (int it = tid; it < nb; it += T) {
            if (pb.m == 3) {                                       // P3P
                uint32_t pos[3];
                clc_acr_sample_t<3>(pb.seed, (uint32_t)(s.iter + it), (uint32_t)s.n_index, pos);
#pragma unroll
                for (int j = 0; j < 3; ++j) samples[it * 3 + j] = (int32_t)(src ? src[pos[j]] : pos[j]);
            } else {                                               // five-point
                uint32_t pos[5];
                clc_acr_sample_t<5>(pb.seed, (uint32_t)(s.iter + it), (uint32_t)s.n_index, pos);
#pragma unroll
                for (int j = 0; j < 5; ++j) samples[it * 5 + j] = (int32_t)(src ? src[pos[j]] : pos[j]);
            }
        }
    }
    // A round that evaluated nothing (the one enqueued ahead of the host's knowledge, after the run has ended) touches NO host memory:
    // the host may already be preparing the next solve in the same pinned block.
    const bool done = B > 0 && s.iter >= s.n_iter;
    if (done && fin.d_res) {
        // the run ends here: the result goes out with this launch (best_inliers was written above by this workgroup)
        __syncthreads();
        acr_finish_block(pb, s, best_inliers, fin, tid, T);
        __threadfence_system();                 // every thread: its stores to the pinned result have left before the word says "done"
    }
    __syncthreads();
    if (tid == 0) {
        *state = s;
        // what the host needs between rounds, in ONE 8-byte word it polls in pinned memory:
        // [63:49] round number, [48] index set switched, [47:40] iterations consumed, [39:20] n_iter, [19:0] iter
        if (h_word && B > 0) {
            const unsigned long long w = ((unsigned long long)((uint32_t)s.rounds & 0x7FFFu) << 49) | ((unsigned long long)(s.index_all ? 0u : 1u) << 48) |
                                         ((unsigned long long)((uint32_t)s.last_batch & 0xFFu) << 40) |
                                         ((unsigned long long)((uint32_t)s.n_iter & 0xFFFFFu) << 20) | (unsigned long long)((uint32_t)s.iter & 0xFFFFFu);
            __hip_atomic_store(h_word, w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- finish: mask, inlier list and the result record, straight into pinned host memory ---------------------------
__global__ __launch_bounds__(256) void acr_finish_kernel(const AcrProblem pb, const AcrState* __restrict__ state,
                                                         const uint32_t* __restrict__ best_inliers, uint8_t* __restrict__ d_mask,
                                                         AcrResult* __restrict__ d_res, uint8_t* __restrict__ h_mask,
                                                         int32_t* __restrict__ h_inliers, AcrResult* __restrict__ h_res)
{
    const AcrState s = *state;
    const AcrFinish fin{ d_mask, d_res, h_mask, h_inliers, h_res };
    acr_finish_block(pb, s, best_inliers, fin, (int)threadIdx.x, (int)blockDim.x);        // launched as ONE workgroup
}

// ---- host side ----------------------------------------------------------------------------------------------------
template <int E>
static hipError_t acr_launch_nfa(const AcrProblem& pb, int B, int P, const double* d_models, AcrHyp* d_hyp, uint32_t* d_sorted,
                                 const AcrState* d_state, hipStream_t stream)
{
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)acr_nfa_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAcrMaxLds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const int T = P / E;
    // LDS is only touched by exchanges that cross waves
    const size_t lds = T > 64 ? (size_t)P * 8 : 0;
    hipLaunchKernelGGL(acr_nfa_kernel<E>, dim3(B * pb.max_models), dim3(T), lds, stream, pb, P, d_models, d_hyp, d_sorted, d_state);
    return hipGetLastError();
}

hipError_t launch_acr_round(const AcrProblem& pb, const double* d_models, AcrHyp* d_hyp, uint32_t* d_sorted, AcrState* d_state,
                            uint32_t* d_best_inliers, uint32_t* d_index_set, int32_t* d_samples, unsigned long long* h_word,
                            hipStream_t stream, int batch_bound, uint8_t* d_mask, AcrResult* d_res, uint8_t* h_mask, int32_t* h_inliers,
                            AcrResult* h_res)
{
    // the grid covers `batch_bound` iterations (an upper bound of the batch the device state will ask for, <= kAcrMaxBatch); the
    // kernels take the round's real batch from the device state.  A tight bound matters: a slot workgroup that has nothing to do
    // still costs its dispatch (512 workgroups of 1024 threads for a 25-iteration reserve round was a third of the nfa launch).
    const int B = batch_bound < 1 ? 1 : (batch_bound > kAcrMaxBatch ? kAcrMaxBatch : batch_bound);
    int P = 64;
    while (P < pb.n) P <<= 1;
        hipError_t e;
    // up to 1024 threads per slot: one element per thread wins while it fits (measured: p50 0.281 -> 0.253 ms at n = 1000
    // against four per thread -- the residual / log10 arithmetic is latency-bound with one wave per SIMD)
    if (P <= 1024) e = acr_launch_nfa<1>(pb, B, P, d_models, d_hyp, d_sorted, d_state, stream);
    else if (P == 2048) e = acr_launch_nfa<2>(pb, B, P, d_models, d_hyp, d_sorted, d_state, stream);
    else if (P == 4096) e = acr_launch_nfa<4>(pb, B, P, d_models, d_hyp, d_sorted, d_state, stream);
    else if (P == 8192) e = acr_launch_nfa<8>(pb, B, P, d_models, d_hyp, d_sorted, d_state, stream);
    else e = acr_launch_nfa<16>(pb, B, P, d_models, d_hyp, d_sorted, d_state, stream);
    if (e != hipSuccess) return e;
    const AcrFinish fin{ d_mask, d_res, h_mask, h_inliers, h_res };
    hipLaunchKernelGGL(acr_select_kernel, dim3(1), dim3(256), 0, stream, pb, d_models, (const AcrHyp*)d_hyp, (const uint32_t*)d_sorted,
                       d_state, d_best_inliers, d_index_set, d_samples, h_word, fin);
    return hipGetLastError();
}

hipError_t launch_acr_finish(const AcrProblem& pb, const AcrState* d_state, const uint32_t* d_best_inliers, uint8_t* d_mask, AcrResult* d_res,
                             uint8_t* h_mask, int32_t* h_inliers, AcrResult* h_res, hipStream_t stream)
{
    hipLaunchKernelGGL(acr_finish_kernel, dim3(1), dim3(256), 0, stream, pb, d_state, d_best_inliers, d_mask, d_res, h_mask, h_inliers, h_res);
    return hipGetLastError();
}

size_t acr_hyp_bytes() { return sizeof(AcrHyp); }

} // namespace clc
