// acransac.hip -- a-contrario RANSAC (AC-RANSAC) on gfx950: the model selection OpenMVG runs for the reference's pose
// and two-view steps,
//     SfM_Localizer::Localize(P3P_KE_CVPR17, ..., {error_max = +inf, max_iteration = 256})   include/coloc/Localizer.hpp:82-93
//     ACRANSAC(ACKernelAdaptorEssential<FivePointSolver, SymmetricEpipolarDistanceError>, ...)  include/coloc/RobustMatcher.hpp:153-171
//     ACRANSAC(ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError, UnnormalizerT>, ...)    :128-151  ('F', round 6)
//     ACRANSAC(ACKernelAdaptor<FourPointSolver, AsymmetricError, UnnormalizerI>, ...)           :188-239  ('H', round 6)
// restated from the published algorithm (Moisan, Moulon, Monasse, IPOL 2012; see oracle/clc_oracle_acr.c for the
// sequential form and what is unpinned).  Per model: residuals over ALL data, sorted with their indices, NFA(k) over
// the k smallest, the model's value = min_k NFA(k); the run keeps the model with the lowest value, and once a
// meaningful one (NFA < 0) exists the remaining 10 % reserve of iterations sample among its inliers.
//
// GPU shape.  The sequential loop evaluates <= 4 (10) models per iteration, one after the other, each with an O(n log n)
// sort on one CPU thread.  Here a ROUND evaluates a batch of B iterations at once:
//   solve   : the existing minimal solvers (p3p_kernel: one problem per four lanes; fivept_kernel: one per wave) on the
//             batch's samples -- samples are a pure function of (seed, iteration, index set), see clc_acr.h;
//   nfa     : one workgroup per model slot: residuals in registers, sort of one 64-bit word per element (residual bits with the
//             element index in the low mantissa bits) in registers / across lanes / through LDS, exact order restored from the
//             exact residual bits, NFA(k) for every k in parallel, min-reduction -> {nfa, k, e_k} and the sorted index list;
//   select  : ONE WAVE replays the SEQUENTIAL semantics over the batch in iteration / solver order (strict '<' improvements,
//             the phase-switch rule), stops at the first iteration that changes the index set -- the iterations after it were
//             sampled speculatively from the old set and are discarded; one workgroup then applies the side effects (device
//             state, best inlier list, index set, the word the host polls, the result record when the run ends).
// The resection path -- and the seven-point / four-point paths, whose solves are as short as a P3P -- runs solve + nfa + the select of
// the PREVIOUS round as ONE launch per round (acr_round_kernel<E, KIND> below), the
// five-point path as two (acr_solve5_kernel: select of the previous round + samples + solve; nfa); several solves of one kind can share
// those launches (blockIdx.y = solve: the *_chains_kernel forms, driven in lockstep by pose_batch.hip).  The host only polls one packed word per round in pinned memory to learn whether
// another round is needed: one round to find the first meaningful model, then one per improvement in the reserve.  Results are identical to the
// sequential oracle: same samples, bit-identical residuals (same operation order, no FMA contraction), a total order on
// (residual, index), and the same portable log10 in the NFA terms.
#include "clc_internal.h"
#include "clc_acr.h"
#include "p3p.h"
#include "fivept_wave.h"
#include "twoview_min.h"

namespace clc {

// -DCLC_ACR_STAMP (experiments only, tools/archive/acr_stamps.py): thread 0 of slot workgroup 0 leaves s_memtime at the phase boundaries of
// acr_round_kernel in a device array that clc_debug_acr_stamps copies out
#if defined(CLC_ACR_STAMP)
__device__ unsigned long long g_acr_stamp[16];
#define ACR_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_acr_stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ACR_STAMP(i) do { } while (0)
#endif

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
        // unormalizeError: resection sqrt(e) / N1(0,0), fundamental / homography sqrt(e) / N2(0,0) -> pixels; essential: the squared
        // pixel distance as it is
        r.error_max = !ok ? 0.0 : (pb.kind != 1 ? sqrt(s.error_max) / pb.norm : s.error_max);
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

// fundamental::kernel::EpipolarDistanceError: squared distance of x2 to the line F x1 (normalised coordinates); operation order of
// oracle/clc_oracle_twoview.c orc_tv_residuals
__device__ __forceinline__ double acr_err_line(const double* __restrict__ f, const double u1, const double v1, const double u2, const double v2)
{
    const double a0 = (f[0] * u1 + f[1] * v1) + f[2];
    const double a1 = (f[3] * u1 + f[4] * v1) + f[5];
    const double a2 = (f[6] * u1 + f[7] * v1) + f[8];
    const double d = (u2 * a0 + v2 * a1) + a2;
    return (d * d) / (a0 * a0 + a1 * a1);
}
// homography::kernel::AsymmetricError: squared transfer error |x2 - H x1|^2 (normalised coordinates)
__device__ __forceinline__ double acr_err_transfer(const double* __restrict__ h, const double u1, const double v1, const double u2, const double v2)
{
    const double a0 = (h[0] * u1 + h[1] * v1) + h[2];
    const double a1 = (h[3] * u1 + h[4] * v1) + h[5];
    const double a2 = (h[6] * u1 + h[7] * v1) + h[8];
    const double du = u2 - a0 / a2, dv = v2 - a1 / a2;
    return du * du + dv * dv;
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
__device__ __forceinline__ double acr_fmin_plain(const double a, const double b) { return b < a ? b : a; }
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
// Wave-wide reductions and the min-scan on DPP moves only (a __shfl is an LDS-crossbar round trip, ~150-300 cycles per dependent
// step when 16 waves share the CU: the replay's scan + three reductions and the NFA reduction were 36 such steps each).  Row steps:
// quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror leave a row's result in all of its 16 lanes; row_bcast:15 / :31 then
// carry row totals upwards, so lane 63 holds the wave's result (readlane).  `old` is what a lane outside the row mask keeps.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int acr_dpp(const int old, const int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double acr_dpp(const double old, const double v)
{
    const uint64_t o = (uint64_t)__double_as_longlong(old), u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)acr_dpp<CTRL, ROW_MASK>((int)(uint32_t)o, (int)(uint32_t)u);
    const uint32_t hi = (uint32_t)acr_dpp<CTRL, ROW_MASK>((int)(uint32_t)(o >> 32), (int)(uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
constexpr int kDppQuad1 = 0xB1, kDppQuad2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppBcast15 = 0x142, kDppBcast31 = 0x143,
              kDppWaveShr1 = 0x138;
__device__ __forceinline__ int acr_wave_min(int v)
{
    int o;
    o = acr_dpp<kDppQuad1>(v, v); v = o < v ? o : v;
    o = acr_dpp<kDppQuad2>(v, v); v = o < v ? o : v;
    o = acr_dpp<kDppHalfMirror>(v, v); v = o < v ? o : v;
    o = acr_dpp<kDppMirror>(v, v); v = o < v ? o : v;
    o = acr_dpp<kDppBcast15, 0xA>(v, v); v = o < v ? o : v;
    o = acr_dpp<kDppBcast31, 0xC>(v, v); v = o < v ? o : v;
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int acr_wave_max(const int v) { return -acr_wave_min(-v); }
// inclusive prefix minimum over the 64 lanes (row_shr:1..3, :4, :8 inside a row, then the row totals), and the exclusive one
template <int SHR>
__device__ __forceinline__ double acr_row_shr(const double ident, const double v) { return acr_dpp<0x110 + SHR>(ident, v); }
__device__ __forceinline__ double acr_wave_scan_min(double v, const double ident /* +inf */, double& excl)
{
    const double t1 = acr_row_shr<1>(ident, v), t2 = acr_row_shr<2>(ident, v), t3 = acr_row_shr<3>(ident, v);
    v = acr_fmin_plain(acr_fmin_plain(v, t1), acr_fmin_plain(t2, t3));
    v = acr_fmin_plain(v, acr_row_shr<4>(ident, v));
    v = acr_fmin_plain(v, acr_row_shr<8>(ident, v));
    v = acr_fmin_plain(v, acr_dpp<kDppBcast15, 0xA>(ident, v));
    v = acr_fmin_plain(v, acr_dpp<kDppBcast31, 0xC>(ident, v));
    excl = acr_dpp<kDppWaveShr1>(ident, v);
    return v;
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
// (Tried: the direction of a step as a constant lane pattern in scalar registers -- two DPP moves, one 64-bit compare into a lane
// mask, a scalar xnor, two selects on the mask: 5 vector instructions instead of 9.  With the stage a run-time value (the general
// network) building the pattern costs more than it saves: sort of a 1 024-element slot 16.0 k -> 24.1 k cycles; with compile-time
// stages (acr_rank_sort) 14.9 k -> 14.1 k, not worth a second form of the step.)

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

// One element per thread (P = T <= 1024): the in-wave stages of the network, then RANK MERGES instead of its cross-wave stages.
// After stage 64 (run ascending in every wave) each wave holds a sorted run of 64; up to four runs are merged at a time by letting
// every element find, by binary search in LDS, how many elements of each sibling run are smaller (keys are distinct: the index is
// part of the key) -- its place in the merged run is the sum.  1 024 elements: 21 in-wave steps + 2 merges (7 and 9 dependent LDS
// reads, three independent searches in flight, 5 workgroup barriers) instead of 21 + 24 in-wave steps and 10 cross-wave exchanges
// with 20 barriers: the sort of a P3P slot went from 16.0 k to 14.6-15.0 k cycles (in-kernel stamps, N = 1000) -- the searches are LDS-throughput
// bound (random 8-byte reads), which is why it is not the 2 x this count suggests.
// (Tried: one spare word per 16 elements so that the power-of-two strides of a binary search step do not share an LDS bank: the
// sort went from 14.6 k to 16.4 k cycles -- the merges are not conflict-bound; the index arithmetic cost more than it saved.)
__device__ __forceinline__ int acr_lower_bound3(const uint64_t* __restrict__ lkey, const int b0, const int b1, const int b2,
                                                const int n_runs, const int L, const uint64_t x)
{
    // number of elements < x in up to three sorted runs of length L (a power of two) starting at elements b0, b1, b2, the searches
    // interleaved
    int p0 = 0, p1 = 0, p2 = 0;
    for (int s = L >> 1; s > 0; s >>= 1) {
        const uint64_t v0 = lkey[b0 + p0 + s - 1], v1 = n_runs > 1 ? lkey[b1 + p1 + s - 1] : 0, v2 = n_runs > 2 ? lkey[b2 + p2 + s - 1] : 0;
        p0 += v0 < x ? s : 0;
        p1 += (n_runs > 1 && v1 < x) ? s : 0;
        p2 += (n_runs > 2 && v2 < x) ? s : 0;
    }
    const uint64_t v0 = lkey[b0 + p0], v1 = n_runs > 1 ? lkey[b1 + p1] : 0, v2 = n_runs > 2 ? lkey[b2 + p2] : 0;
    p0 += v0 < x ? 1 : 0;
    p1 += (n_runs > 1 && v1 < x) ? 1 : 0;
    p2 += (n_runs > 2 && v2 < x) ? 1 : 0;
    return p0 + p1 + p2;
}
__device__ __forceinline__ void acr_rank_sort(double& c, const int P, const int tid, const int T, uint64_t* lkey)
{
    double cc[1] = { c };
    // stages 2 .. 32 of the bitonic network (direction by the element's position), stage 64 ascending everywhere
    acr_step_wave<1, 1>(cc, 2, tid);
    acr_step_wave<1, 2>(cc, 4, tid); acr_step_wave<1, 1>(cc, 4, tid);
    acr_step_wave<1, 4>(cc, 8, tid); acr_step_wave<1, 2>(cc, 8, tid); acr_step_wave<1, 1>(cc, 8, tid);
    acr_step_wave<1, 8>(cc, 16, tid); acr_step_wave<1, 4>(cc, 16, tid); acr_step_wave<1, 2>(cc, 16, tid); acr_step_wave<1, 1>(cc, 16, tid);
    acr_step_wave<1, 16>(cc, 32, tid); acr_step_wave<1, 8>(cc, 32, tid); acr_step_wave<1, 4>(cc, 32, tid); acr_step_wave<1, 2>(cc, 32, tid);
    acr_step_wave<1, 1>(cc, 32, tid);
    const int up = 2048;                                                  // (tid & up) == 0 for every thread: ascending
    acr_step_wave<1, 32>(cc, up, tid); acr_step_wave<1, 16>(cc, up, tid); acr_step_wave<1, 8>(cc, up, tid); acr_step_wave<1, 4>(cc, up, tid);
    acr_step_wave<1, 2>(cc, up, tid); acr_step_wave<1, 1>(cc, up, tid);
    uint64_t x = (uint64_t)__double_as_longlong(cc[0]);
    if (T > 64) {
        lkey[tid] = x;
        __syncthreads();
        for (int L = 64; L < P;) {
            const int F = P / L < 4 ? P / L : 4, G = L * F;               // F runs of length L -> one of length G
            const int base = tid & ~(G - 1), mine = (tid & (G - 1)) / L;  // (uniform over a wave: L >= 64)
            const int to = base + (tid & (L - 1)) +
                           acr_lower_bound3(lkey, base + ((mine + 1) % F) * L, base + ((mine + 2) % F) * L, base + ((mine + 3) % F) * L, F - 1, L, x);
            __syncthreads();                                              // every search has read the runs
            lkey[to] = x;
            __syncthreads();
            x = lkey[tid];
            L = G;
        }
    }
    c = __longlong_as_double((long long)x);
}

// What a thread's part of a slot needs from memory that does not depend on the model: its E correspondences and the two table
// entries of its E positions.  acr_round_kernel loads them before anything else, so that they arrive while wave 0 is still
// replaying the previous round and solving the sample (resection only; PRE = false: loaded where they are used).
template <int E>
struct AcrPre { double a[3 * E], b[2 * E]; float cn[E], ck[E]; };
template <int E>
__device__ __forceinline__ void acr_prefetch(const AcrProblem& pb, const int tid, AcrPre<E>& q)
{
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid * E + e < pb.n ? tid * E + e : 0, kk = tid * E + e + 1 <= pb.n ? tid * E + e + 1 : 0;
        q.a[3 * e] = pb.a[3 * i]; q.a[3 * e + 1] = pb.a[3 * i + 1]; q.a[3 * e + 2] = pb.a[3 * i + 2];
        q.b[2 * e] = pb.b[2 * i]; q.b[2 * e + 1] = pb.b[2 * i + 1];
        q.cn[e] = pb.logc_n[kk]; q.ck[e] = pb.logc_k[kk];
    }
}

// (NFA, its residual, its k) -- lexicographic minimum on (NFA, k) -- and a count that is summed: kept as four scalars, not a struct
// (the compiler turns selects between two structs into selects between their stack addresses and the structs stay in scratch)
__device__ __forceinline__ void acr_best_merge(double& v, double& e, int& k, int& cnt, const double ov, const double oe, const int ok, const int oc)
{
    const bool take = ov < v || (ov == v && ok < k);
    v = take ? ov : v; e = take ? oe : e; k = take ? ok : k;
    cnt += oc;
}
template <int CTRL>
__device__ __forceinline__ void acr_best_step(double& v, double& e, int& k, int& cnt)
{
    const double ov = acr_dpp<CTRL>(v, v), oe = acr_dpp<CTRL>(e, e);
    const int ok = acr_dpp<CTRL>(k, k), oc = acr_dpp<CTRL>(cnt, cnt);
    acr_best_merge(v, e, k, cnt, ov, oe, ok, oc);
}
__device__ __forceinline__ void acr_best_rows(double& v, double& e, int& k, int& cnt)     // every lane of a 16-lane row ends up with the row's result
{
    acr_best_step<kDppQuad1>(v, e, k, cnt);
    acr_best_step<kDppQuad2>(v, e, k, cnt);
    acr_best_step<kDppHalfMirror>(v, e, k, cnt);
    acr_best_step<kDppMirror>(v, e, k, cnt);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void acr_best_bcast(double& v, double& e, int& k, int& cnt)    // rows outside the mask merge with (themselves, 0)
{
    const double ov = acr_dpp<CTRL, ROW_MASK>(v, v), oe = acr_dpp<CTRL, ROW_MASK>(e, e);
    const int ok = acr_dpp<CTRL, ROW_MASK>(k, k), oc = acr_dpp<CTRL, ROW_MASK>(0, cnt);
    acr_best_merge(v, e, k, cnt, ov, oe, ok, oc);
}

// One model slot, one workgroup of T threads: `model` (global memory or LDS) -> *hyp_slot and the sorted index list out_idx[0..n).
template <int E, bool PRE = false>
__device__ __forceinline__ void acr_nfa_body(const AcrProblem& pb, const int P /* = T * E, power of two >= n */, const double* model,
                                             AcrHyp* __restrict__ hyp_slot, uint32_t* __restrict__ out_idx, const int tid, const int T,
                                             uint64_t* lkey /* [e][tid] staging of the cross-wave exchanges (one word per element) */,
                                             const AcrPre<PRE ? E : 1>* pre = nullptr)
{
    // E <= 8: the exact residual bits stay in a second LDS array [P] indexed by element, so that after the sort an element's exact
    // key is one LDS read instead of a gather of its correspondence + the residual arithmetic again (2.3 -> ~0.5 us of the launch)
    constexpr bool kKeep = E <= 8;
    uint64_t* lres = lkey + (T > 64 ? P : 0);
    const int n = pb.n;
    uint32_t* lidx = nullptr;
    __shared__ double s_nfa[1024 / 64], s_ek[1024 / 64];
    __shared__ int s_k[1024 / 64], s_cnt[1024 / 64];
    __shared__ uint64_t s_edge_key[1024 / 64];
    __shared__ uint32_t s_edge_idx[1024 / 64];
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    // an empty slot (the solver marks it with NaNs) never improves anything
    if (model[0] != model[0]) {
        if (tid == 0) { hyp_slot->nfa = inf; hyp_slot->e_k = 0.0; hyp_slot->k = 0; hyp_slot->n_le = 0; }
        return;
    }
    auto residual = [&](const int i) -> double {
        if (pb.kind == 0) return acr_err_resection(model, pb.K1v, pb.norm, pb.a[3 * i], pb.a[3 * i + 1], pb.a[3 * i + 2], pb.b[2 * i], pb.b[2 * i + 1]);
        const double u1 = pb.a[2 * i], v1 = pb.a[2 * i + 1], u2 = pb.b[2 * i], v2 = pb.b[2 * i + 1];
        return pb.kind == 1 ? acr_err_epipolar(model, u1, v1, u2, v2) : (pb.kind == 2 ? acr_err_line(model, u1, v1, u2, v2) : acr_err_transfer(model, u1, v1, u2, v2));
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
            double r;
            if constexpr (PRE) r = acr_err_resection(model, pb.K1v, pb.norm, pre->a[3 * e], pre->a[3 * e + 1], pre->a[3 * e + 2], pre->b[2 * e], pre->b[2 * e + 1]);
            else r = residual(i);
            cnt += r <= pb.max_threshold ? 1 : 0;
            const uint64_t rb = (uint64_t)__double_as_longlong(r);
            bits = rb < kMaxFinite ? rb : kMaxFinite;
            if (kKeep) lres[i] = rb;
        }
        c[e] = __longlong_as_double((long long)((bits & ~kIdxMask) | (uint64_t)i));
    }
    ACR_STAMP(6);
    if constexpr (E == 1) acr_rank_sort(c[0], P, tid, T, lkey);
    else acr_bitonic<E, false>(key, idx, c, P, tid, T, lkey, lidx);
    ACR_STAMP(7);
    // exact keys of the elements as they stand now, then the neighbour check
    if (kKeep && T > 64) __syncthreads();                                 // (every thread's lres entry is written)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)((uint64_t)__double_as_longlong(c[e]) & kIdxMask);
        if ((int)i < n) { idx[e] = i; key[e] = kKeep ? lres[i] : (uint64_t)__double_as_longlong(residual((int)i)); }
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
    ACR_STAMP(8);
    // NFA(k) of this thread's own positions k = tid E + e + 1, for m + 1 <= k <= n and e_(k) <= max_threshold; strict '<'
    // keeps the first k
    double best = inf, bek = 0.0;
    int bk = pb.m;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int kk = tid * E + e + 1;
        if (kk <= n) out_idx[kk - 1] = idx[e];
        if (kk > pb.m && kk <= n) {
            const double r = __longlong_as_double((long long)key[e]);
            if (r <= pb.max_threshold) {
                float cn, ck;
                if constexpr (PRE) { cn = pre->cn[e]; ck = pre->ck[e]; }
                else { cn = pb.logc_n[kk]; ck = pb.logc_k[kk]; }
                const double v = clc_acr_nfa(pb.loge0, pb.logalpha0, pb.mult, r, kk, pb.m, cn, ck);
                if (v < best) { best = v; bk = kk; bek = r; }
            }
        }
    }
    ACR_STAMP(13);
    // lexicographic minimum of (NFA, k) and the sum of cnt: over the wave on DPP moves (result in lane 63), across the waves through
    // LDS, where the first row of wave 0 reduces the <= 16 wave results the same way
    acr_best_rows(best, bek, bk, cnt);
    acr_best_bcast<kDppBcast15, 0xA>(best, bek, bk, cnt);
    acr_best_bcast<kDppBcast31, 0xC>(best, bek, bk, cnt);
    ACR_STAMP(14);
    if ((tid & 63) == 63) { s_nfa[tid >> 6] = best; s_k[tid >> 6] = bk; s_ek[tid >> 6] = bek; s_cnt[tid >> 6] = cnt; }
    __syncthreads();
    ACR_STAMP(15);
    if (tid < 16) {
        const bool have = tid < (T + 63) / 64;
        double wv = have ? s_nfa[tid] : inf, we = have ? s_ek[tid] : 0.0;
        int wk = have ? s_k[tid] : 0x7fffffff, wc = have ? s_cnt[tid] : 0;
        acr_best_rows(wv, we, wk, wc);
        if (tid == 0) {
            hyp_slot->nfa = wv;
            hyp_slot->k = wk;
            hyp_slot->e_k = we;
            hyp_slot->n_le = wc;
        }
    }
    ACR_STAMP(9);
}

// models from memory (the five-point path: fivept_kernel has written them)
template <int E>
__global__ __launch_bounds__(1024) void acr_nfa_kernel(const AcrProblem pb, const int P, const double* __restrict__ models,
                                                       AcrHyp* __restrict__ hyp, uint32_t* __restrict__ sorted_idx,
                                                       const AcrState* __restrict__ state)
{
    extern __shared__ unsigned char acr_lds[];
    const int slot = blockIdx.x;
    if (slot >= state->cur_batch * pb.max_models) return;          // the grid covers the largest batch; this round is smaller
    acr_nfa_body<E>(pb, P, models + (size_t)slot * pb.model_doubles, hyp + slot, sorted_idx + (size_t)slot * pb.n, (int)threadIdx.x,
                    (int)blockDim.x, reinterpret_cast<uint64_t*>(acr_lds));
}

// the nfa launch of a two-view round for up to kMaxBatch solves at once (blockIdx.y = chain): the copies of launch parity `par`
template <int E>
__global__ __launch_bounds__(1024) void acr_nfa_chains_kernel(const AcrChains chains, const int par, const int P)
{
    extern __shared__ unsigned char acr_lds[];
    constexpr int kSlots = kAcrMaxBatch * 10;
    const AcrChain& ch = chains.c[blockIdx.y];
    const AcrProblem& pb = ch.pb;
    const int slot = blockIdx.x;
    const AcrState* state = ch.states + par;                         // what this round's keeper has just written: cur_batch = this round's batch
    if (slot >= state->cur_batch * pb.max_models) return;
    acr_nfa_body<E>(pb, P, ch.models + ((size_t)par * kSlots + slot) * pb.model_doubles, ch.hyps + (size_t)par * kSlots + slot,
                    ch.sorted + ((size_t)par * kSlots + slot) * pb.n, (int)threadIdx.x, (int)blockDim.x, reinterpret_cast<uint64_t*>(acr_lds));
}

// ---- select: the sequential semantics over one batch -------------------------------------------------------------
// The loop being replayed is sequential (strict '<' improvements in iteration / solver order, the first iteration that
// switches the index set ends the batch), but everything in it is a prefix operation over the <= 1280 model slots: the
// running minimum is a prefix min, "improved" compares a slot with the prefix before it, the batch ends at the FIRST
// iteration whose condition holds.  ONE WAVE does it, two iterations (2 M consecutive slots) per lane: a lane-local
// pass, one min-scan and three reductions across the 64 lanes -- no LDS, no workgroup barrier, so that every workgroup
// of acr_round_kernel can afford to redo it for itself (below).
struct AcrCore {           // AcrState without the model: what the replay reads and updates (registers, uniform over the wave)
    double min_nfa, error_max;
    int32_t n_inliers, best_iter, iter, n_iter, reserve, n_index, index_all, ac_mode, rounds, last_batch, cur_batch, grow, rounds_eval, evaluated;
};
struct AcrPick {           // what the replay of one batch leaves besides the updated state
    int32_t best_h;        // slot of the model the sequential loop ends up with, -1: no improvement in this batch
    int32_t copy_index;    // the index set becomes the inliers of the current best model
    int32_t batch;         // iterations the replayed round had evaluated (0: nothing to replay)
};
__device__ __forceinline__ AcrCore acr_core_load(const AcrState* __restrict__ st)
{
    AcrCore c;
    c.min_nfa = st->min_nfa; c.error_max = st->error_max; c.n_inliers = st->n_inliers; c.best_iter = st->best_iter; c.iter = st->iter;
    c.n_iter = st->n_iter; c.reserve = st->reserve; c.n_index = st->n_index; c.index_all = st->index_all; c.ac_mode = st->ac_mode;
    c.rounds = st->rounds; c.last_batch = st->last_batch; c.cur_batch = st->cur_batch; c.grow = st->grow; c.rounds_eval = st->rounds_eval;
    c.evaluated = st->evaluated;
    return c;
}
__device__ __forceinline__ void acr_core_store(AcrState& st, const AcrCore& c)
{
    st.min_nfa = c.min_nfa; st.error_max = c.error_max; st.n_inliers = c.n_inliers; st.best_iter = c.best_iter; st.iter = c.iter;
    st.n_iter = c.n_iter; st.reserve = c.reserve; st.n_index = c.n_index; st.index_all = c.index_all; st.ac_mode = c.ac_mode;
    st.rounds = c.rounds; st.last_batch = c.last_batch; st.cur_batch = c.cur_batch; st.grow = c.grow; st.rounds_eval = c.rounds_eval;
    st.evaluated = c.evaluated;
}

__device__ __forceinline__ double acr_readlane(const double v, const int lane /* uniform */)
{
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
// s: the state the batch was drawn from (in) -> the state after it (out); hyp: the batch's slots.  Called by one full wave.
template <int M>
__device__ __forceinline__ AcrPick acr_select_wave(const AcrProblem& pb, AcrCore& s, const AcrHyp* __restrict__ hyp, const int lane)
{
    static_assert(kAcrMaxBatch == 128, "two iterations per lane");
    constexpr int C = 2 * M;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    const int B = s.cur_batch, total = B * M, h0 = lane * C;
    // the one memory round trip: this lane's slots, whole records (k and e_k of the winner come out of registers)
    double val[C], ek[C];
    int nle[C], kk[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        // (addresses do not depend on the state: both loads are in flight together; the slot array always has 128 M records)
        const bool in = h0 + c < total;
        const AcrHyp* hp = hyp + h0 + c;
        const double v = hp->nfa;
        const int nl = hp->n_le;
        ek[c] = hp->e_k;
        kk[c] = hp->k;
        val[c] = in ? v : inf;
        nle[c] = in ? nl : 0;
    }
#if defined(CLC_ACR_STAMP)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ACR_STAMP(11);
#endif
    // 1. upper-bound mode gate: slots before the first model with more than 2.5 m residuals under the bound are ignored
    int first_on = 0;
    if (!s.ac_mode) {
        int first_gate = 0x7fffffff;
#pragma unroll
        for (int c = C - 1; c >= 0; --c) if ((double)nle[c] > 2.5 * (double)pb.m && h0 + c < total) first_gate = h0 + c;
        first_on = acr_wave_min(first_gate);
    }
    // 2. prefix minimum of the slot values, seeded with the minimum of the previous rounds
    double lm = inf;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        if (h0 + c < first_on) val[c] = inf;
        lm = val[c] < lm ? val[c] : lm;
    }
    double run;                                                     // minimum over the lanes before this one
    acr_wave_scan_min(lm, inf, run);
    run = s.min_nfa < run ? s.min_nfa : run;
    bool imp[C];
    double pre_last[2];                                            // prefix min INCLUDING the last slot of each of the lane's iterations
#pragma unroll
    for (int c = 0; c < C; ++c) {
        imp[c] = h0 + c < total && val[c] < run;                   // strict: an equal value does not replace the earlier model
        run = val[c] < run ? val[c] : run;
        if (c % M == M - 1) pre_last[c / M] = run;
    }
    // 3. the first iteration that ends the batch
    int ev = B;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int it = lane * 2 + j;
        bool better = false;
#pragma unroll
        for (int k = 0; k < M; ++k) better = better || imp[j * M + k];
        const int cur = s.iter + it;
        if (it < B && ((better && pre_last[j] < 0.0) || (cur + 1 == s.n_iter && s.reserve)) && it < ev) ev = it;
    }
    const int event_it = acr_wave_min(ev), consumed = event_it < B ? event_it + 1 : B;
    // 4. the last improvement among the consumed slots is the model the sequential loop ends up with
    int last_imp = -1;
#pragma unroll
    for (int c = 0; c < C; ++c) if (h0 + c < consumed * M && imp[c]) last_imp = h0 + c;
    const int best_h = acr_wave_max(last_imp);
    ACR_STAMP(12);
    AcrPick pick;
    pick.best_h = best_h;
    pick.copy_index = 0;
    pick.batch = B;
    if (!s.ac_mode && first_on < consumed * M) s.ac_mode = 1;
    if (best_h >= 0) {
        const int owner = best_h / C, cb = best_h % C;
        double v = val[0], e = ek[0];
        int k = kk[0];
#pragma unroll
        for (int c = 1; c < C; ++c) if (c == cb) { v = val[c]; e = ek[c]; k = kk[c]; }
        s.min_nfa = acr_readlane(v, owner);
        s.error_max = acr_readlane(e, owner);
        s.n_inliers = __builtin_amdgcn_readlane(k, owner);
        s.best_iter = s.iter + best_h / M;
    }
    if (event_it < B) {
        const int cur = s.iter + event_it;
        if (s.n_inliers == 0) { s.n_iter++; s.reserve--; }
        else {
            pick.copy_index = 1;
            s.n_index = s.n_inliers;
            s.index_all = 0;
            if (s.reserve) { s.n_iter = cur + 1 + s.reserve; s.reserve = 0; }
        }
    }
    s.iter += consumed;
    if (B > 0) { s.rounds += 1; s.rounds_eval += 1; }            // a launch that found nothing to replay is not a round
    s.last_batch = consumed;
    // next round: while nothing has happened look further ahead per round; after an event the whole reserve goes in one
    if (event_it < B || !s.index_all) s.grow = pb.batch_cap;
    else s.grow = s.grow * 2 > pb.batch_cap ? pb.batch_cap : s.grow * 2;
    const int remaining = s.n_iter - s.iter;
    s.cur_batch = remaining < s.grow ? (remaining > 0 ? remaining : 0) : s.grow;
    return pick;
}

// the list the NEXT batch's sample positions map through, as the index set stands after the replay: read from where it
// came from, not from the copy the keeper is writing at the same time (nullptr: positions are indices)
__device__ __forceinline__ const uint32_t* acr_sample_source(const AcrCore& s, const AcrPick& pick, const uint32_t* win,
                                                             const uint32_t* best_inliers, const uint32_t* index_set)
{
    return s.index_all ? nullptr : (pick.copy_index ? (win ? win : best_inliers) : index_set);
}

// The side effects of a replayed batch, by ONE workgroup (all T threads): winner's sorted index list -> best_inliers and (on a switch)
// the index set, the model, the state record, the word the host polls -- and, when the run ends here, mask / inlier list / result
// record (system-scope fence before the word).  A launch that replayed nothing (batch == 0) only carries the state forward and
// touches NO host memory: the host may already be preparing the next solve in the same pinned block.
__device__ __forceinline__ void acr_keep(const AcrProblem& pb, const AcrCore& c, const AcrPick& pick, const double* __restrict__ models,
                                         const uint32_t* __restrict__ sorted_idx, const AcrState* st_in, AcrState* st_out,
                                         uint32_t* __restrict__ best_inliers, uint32_t* __restrict__ index_set,
                                         unsigned long long* __restrict__ h_word, const AcrFinish& fin, AcrState& s_full /* LDS */,
                                         const int tid, const int T)
{
    const int best_h = pick.best_h, n_inl = c.n_inliers;
    if (tid < 18) s_full.model[tid] = best_h >= 0 && tid < pb.model_doubles ? models[(size_t)best_h * pb.model_doubles + tid] : st_in->model[tid];
    if (tid == 0) acr_core_store(s_full, c);
    const uint32_t* win = best_h >= 0 ? sorted_idx + (size_t)best_h * pb.n : nullptr;
    if (win)
        for (int i = tid; i < n_inl; i += T) {
            const uint32_t v = win[i];
            best_inliers[i] = v;
            if (pick.copy_index) index_set[i] = v;
        }
    else if (pick.copy_index)                           // vec_index = vec_inliers of a model found in an earlier round
        for (int i = tid; i < n_inl; i += T) index_set[i] = best_inliers[i];
    __syncthreads();                                    // s_full is whole, best_inliers is written
    const bool done = pick.batch > 0 && c.iter >= c.n_iter;
    if (done && fin.d_res) {
        // the run ends here: the result goes out with this launch
        acr_finish_block(pb, s_full, best_inliers, fin, tid, T);
        __threadfence_system();                 // every thread: its stores to the pinned result have left before the word says "done"
    }
    __syncthreads();
    if (tid == 0) {
        *st_out = s_full;
        // what the host needs between rounds, in ONE 8-byte word it polls in pinned memory:
        // [63:49] round number, [48] index set switched, [47:40] iterations consumed, [39:20] n_iter, [19:0] iter
        if (h_word && pick.batch > 0) {
            const unsigned long long w = ((unsigned long long)((uint32_t)c.rounds & 0x7FFFu) << 49) | ((unsigned long long)(c.index_all ? 0u : 1u) << 48) |
                                         ((unsigned long long)((uint32_t)c.last_batch & 0xFFu) << 40) |
                                         ((unsigned long long)((uint32_t)c.n_iter & 0xFFFFFu) << 20) | (unsigned long long)((uint32_t)c.iter & 0xFFFFFu);
            __hip_atomic_store(h_word, w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- the resection round in ONE launch ------------------------------------------------------------------------------
// A P3P round used to be three dependent launches (solve 5.8 us, nfa 15 us, select 7.7 us in the kernel trace) of which ~4.4 us
// EACH is the launch boundary itself (an empty launch in the same chain measures 4.0-4.5 us): 13 of a round's 28 us.  Here a round
// is one launch, and no workgroup waits for another inside it:
//   * every workgroup first REPLAYS THE PREVIOUS ROUND for itself (acr_select_wave on wave 0: deterministic, ~1 memory round trip,
//     the inputs are the previous launch's outputs) -- that tells it this round's batch, iteration numbers and index set;
//   * slot workgroup (it, root): draws the sample of iteration `it`, solves root `root` of its P3P problem (the shared, not-inlined
//     p3p_sample_root: same bits as p3p_kernel), then residuals / sort / NFA as before;
//   * the last workgroup of the grid is the KEEPER: it alone applies the previous round's side effects (acr_keep).
// Everything a launch writes that a workgroup of the SAME launch might still read lives in two copies indexed by launch parity
// (state, slots, sorted lists, models); best_inliers / index_set are written by the keeper only in the cases in which the slot
// workgroups read the other list (acr_sample_source).  The word of round r therefore comes out of launch r + 1 -- early in it, so
// the host has launch r + 2 enqueued long before r + 1 ends, and the launch that reports "done" is the last one in the stream.
// The same one-launch round serves the two cheap two-view solvers (round 6: 'F' seven points, 'H' four points -- twoview_min.h): a
// slot workgroup solves its iteration's sample on thread 0 (all the slots of an iteration solve the same sample and keep their own
// root), everything else is the resection round.  KIND = AcrProblem::kind; slots and models keep the resection's strides (4 slots per
// iteration between the parity copies, 12 doubles per model) so that the host lays one workspace out for all three.
template <int KIND> struct AcrKind;
template <> struct AcrKind<0> { static constexpr int m = 3, M = 4; };
template <> struct AcrKind<2> { static constexpr int m = 7, M = 3; };
template <> struct AcrKind<3> { static constexpr int m = 4, M = 1; };

// the model of root `root` of the sample {id[0..m)} of a seven-point / four-point problem -> out[0..12) (9 used; NaN: no such root)
template <int KIND>
__device__ __forceinline__ void acr_twoview_sample_root(const double* __restrict__ x1, const double* __restrict__ x2, const int (&id)[AcrKind<KIND>::m],
                                                        const int N, const int root, double* out)
{
    constexpr int m = AcrKind<KIND>::m;
    double a[m][2], b[m][2];
    bool ok = true;
#pragma unroll
    for (int p = 0; p < m; ++p) {
        int i = id[p];
        if (i < 0 || i >= N) { ok = false; i = 0; }
        a[p][0] = x1[2 * i]; a[p][1] = x1[2 * i + 1];
        b[p][0] = x2[2 * i]; b[p][1] = x2[2 * i + 1];
    }
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    double M9[9];
    bool have;
    if constexpr (KIND == 2) {
        double F[3][9];
        const int nr = tv::seven_point(a, b, F);
        have = ok && root < nr;
#pragma unroll
        for (int e = 0; e < 9; ++e) M9[e] = root == 0 ? F[0][e] : (root == 1 ? F[1][e] : F[2][e]);
    } else {
        have = ok && tv::four_point(a, b, M9) > 0 && root == 0;
    }
    have = have && M9[0] == M9[0];
#pragma unroll
    for (int e = 0; e < 12; ++e) out[e] = have ? (e < 9 ? M9[e] : 0.0) : qnan;
}

template <int E, int KIND>
__device__ __forceinline__ void acr_round_body(AcrState* __restrict__ states /* [2] */, AcrHyp* __restrict__ hyps /* [2][slots] */,
                                               const int par, const int P /* = blockDim.x * E */, const AcrProblem& pb,
                                               uint32_t* __restrict__ sorted /* [2][slots * n] */,
                                               double* __restrict__ models /* [2][slots * 12] */,
                                               uint32_t* __restrict__ best_inliers, uint32_t* __restrict__ index_set,
                                               unsigned long long* __restrict__ h_word, const AcrFinish& fin)
{
    extern __shared__ unsigned char acr_lds[];
    constexpr int kSlots = kAcrMaxBatch * 4;
    __shared__ AcrCore s_core;
    __shared__ AcrPick s_pick;
    __shared__ AcrState s_full;
    __shared__ double s_model[12];
    const int tid = threadIdx.x, T = blockDim.x, slot = blockIdx.x;
    const bool keeper = blockIdx.x == gridDim.x - 1;
#if defined(CLC_ACR_STAMP)
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    unsigned long long t_replay = 0;
#endif
    constexpr int m = AcrKind<KIND>::m, M = AcrKind<KIND>::M;
    constexpr bool kPre = KIND == 0 && E <= 2;                      // (more elements per thread: the registers are worth more than the latency)
    AcrPre<kPre ? E : 1> pre;
    if (kPre && !keeper) acr_prefetch<kPre ? E : 1>(pb, tid, pre);
    const AcrState* st_in = states + (par ^ 1);
    const AcrHyp* hyp_in = hyps + (size_t)(par ^ 1) * kSlots;
    const uint32_t* sorted_in = sorted + (size_t)(par ^ 1) * kSlots * pb.n;
    const double* models_in = models + (size_t)(par ^ 1) * kSlots * 12;
    if (tid < 64) {
        AcrCore c = acr_core_load(st_in);
        AcrPick pick{ -1, 0, 0 };
        if (c.evaluated && c.cur_batch > 0) pick = acr_select_wave<M>(pb, c, hyp_in, tid);   // else: the first launch of a run, or a launch after its end
        c.evaluated = 1;
        if (tid == 0) { s_core = c; s_pick = pick; }
#if defined(CLC_ACR_STAMP)
        t_replay = __builtin_amdgcn_s_memtime();
#endif
    }
    __syncthreads();
    const AcrCore s = s_core;
    const AcrPick pick = s_pick;
    if (keeper) {
        acr_keep(pb, s, pick, models_in, sorted_in, st_in, states + par, best_inliers, index_set, h_word, fin, s_full, tid, T);
        return;
    }
    if (slot >= s.cur_batch * M) return;                            // the grid covers an upper bound of the batch
    if (tid == 0) {
#if defined(CLC_ACR_STAMP)
        if (blockIdx.x == 0) { g_acr_stamp[0] = t_start; g_acr_stamp[1] = t_replay; }
#endif
        ACR_STAMP(2);
        const uint32_t* win = pick.best_h >= 0 ? sorted_in + (size_t)pick.best_h * pb.n : nullptr;
        const uint32_t* src = acr_sample_source(s, pick, win, best_inliers, index_set);
        uint32_t pos[m];
        clc_acr_sample_t<m>(pb.seed, (uint32_t)(s.iter + slot / M), (uint32_t)s.n_index, pos);
        int id[m];
#pragma unroll
        for (int j = 0; j < m; ++j) id[j] = (int)(src ? src[pos[j]] : pos[j]);
#if defined(CLC_ACR_STAMP)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        ACR_STAMP(3);
        if constexpr (KIND == 0) p3p_sample_root(pb.a, pb.b, pb.K1v, id[0], id[1], id[2], pb.n, slot & 3, s_model);
        else acr_twoview_sample_root<KIND>(pb.a, pb.b, id, pb.n, slot % M, s_model);
        ACR_STAMP(4);
        double* mo = models + ((size_t)par * kSlots + slot) * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) mo[e] = s_model[e];
    }
    __syncthreads();
    ACR_STAMP(5);
    acr_nfa_body<E, kPre>(pb, P, s_model, hyps + (size_t)par * kSlots + slot, sorted + ((size_t)par * kSlots + slot) * pb.n, tid, T,
                          reinterpret_cast<uint64_t*>(acr_lds), &pre);
}

template <int E, int KIND>
__global__ __launch_bounds__(1024) void acr_round_kernel(AcrState* __restrict__ states /* [2] */, AcrHyp* __restrict__ hyps /* [2][slots] */,
                                                         const int par, const int P /* = blockDim.x * E */, const AcrProblem pb,
                                                         uint32_t* __restrict__ sorted /* [2][slots * n] */,
                                                         double* __restrict__ models /* [2][slots * 12] */,
                                                         uint32_t* __restrict__ best_inliers, uint32_t* __restrict__ index_set,
                                                         unsigned long long* __restrict__ h_word, const AcrFinish fin)
{
    acr_round_body<E, KIND>(states, hyps, par, P, pb, sorted, models, best_inliers, index_set, h_word, fin);
}
// the same round for up to kMaxBatch solves at once: blockIdx.y = chain (same code, same bits: the file is compiled without contraction)
template <int E, int KIND>
__global__ __launch_bounds__(1024) void acr_round_chains_kernel(const AcrChains chains, const int par, const int P)
{
    const AcrChain& ch = chains.c[blockIdx.y];
    acr_round_body<E, KIND>(ch.states, ch.hyps, par, P, ch.pb, ch.sorted, ch.models, ch.best_inliers, ch.index_set, ch.h_word, ch.fin);
}

// The seven-point / four-point models of caller-chosen samples (clc_two_view_minimal: the hypothesis generator the tests hand to the
// sequential oracle, as p3p_kernel / fivept_kernel are for the other two kinds): one thread per (sample, root), the body the round runs.
template <int KIND>
__global__ __launch_bounds__(64) void twoview_minimal_kernel(const double* __restrict__ x1, const double* __restrict__ x2, const int N,
                                                             const int32_t* __restrict__ samples, const int S, double* __restrict__ out /* S x M x 9 */)
{
    constexpr int m = AcrKind<KIND>::m, M = AcrKind<KIND>::M;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * M) return;
    const int sidx = t / M, root = t % M;
    int id[m];
#pragma unroll
    for (int j = 0; j < m; ++j) id[j] = samples[sidx * m + j];
    double mo[12];
    acr_twoview_sample_root<KIND>(x1, x2, id, N, root, mo);
#pragma unroll
    for (int e = 0; e < 9; ++e) out[(size_t)t * 9 + e] = mo[e];
}
hipError_t launch_twoview_minimal(int kind, const double* d_x1, const double* d_x2, int N, const int32_t* d_samples, int S, double* d_out, hipStream_t stream)
{
    if (S <= 0) return hipSuccess;
    if (kind == 2) hipLaunchKernelGGL(twoview_minimal_kernel<2>, dim3((S * 3 + 63) / 64), dim3(64), 0, stream, d_x1, d_x2, N, d_samples, S, d_out);
    else if (kind == 3) hipLaunchKernelGGL(twoview_minimal_kernel<3>, dim3((S + 63) / 64), dim3(64), 0, stream, d_x1, d_x2, N, d_samples, S, d_out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- the five-point round in TWO launches (round 5) ---------------------------------------------------------------------------
// The two-view round was solve -> nfa -> select: three launches, 39 + 11 + 6.5 us of kernels and three launch boundaries of ~4.4 us.
// The select launch goes the way it went for resection: every solver workgroup replays the previous round for itself (one wave, one
// memory round trip) to learn this round's batch, iteration numbers and index set, draws its own sample and solves it; the last
// workgroup of the grid is the keeper (side effects of the replayed round, the word the host polls).  The nfa launch follows -- the
// solve is one WAVE per iteration (ten models), the nfa one workgroup per MODEL, and a slot workgroup cannot start before its solve
// has ended, so the two stay two launches; state, slots, sorted lists and models live in two copies indexed by launch parity exactly
// as for resection.  The solve is fpw::models_of_sample, the one not-inlined body fivept_kernel runs: same bits as the hypotheses the
// tests hand to the sequential oracle.
__device__ __forceinline__ void acr_solve5_body(AcrState* __restrict__ states /* [2] */, AcrHyp* __restrict__ hyps /* [2][slots] */,
                                                const int par, const AcrProblem& pb, uint32_t* __restrict__ sorted /* [2][slots * n] */,
                                                double* __restrict__ models /* [2][slots * 18] */,
                                                uint32_t* __restrict__ best_inliers, uint32_t* __restrict__ index_set,
                                                unsigned long long* __restrict__ h_word, const AcrFinish& fin)
{
    constexpr int kSlots = kAcrMaxBatch * 10;
    __shared__ AcrCore s_core;
    __shared__ AcrPick s_pick;
    __shared__ AcrState s_full;
    const int tid = threadIdx.x, it = blockIdx.x;
    const bool keeper = blockIdx.x == gridDim.x - 1;
    const AcrState* st_in = states + (par ^ 1);
    const AcrHyp* hyp_in = hyps + (size_t)(par ^ 1) * kSlots;
    const uint32_t* sorted_in = sorted + (size_t)(par ^ 1) * kSlots * pb.n;
    const double* models_in = models + (size_t)(par ^ 1) * kSlots * 18;
    {
        AcrCore c = acr_core_load(st_in);
        AcrPick pick{ -1, 0, 0 };
        if (c.evaluated && c.cur_batch > 0) pick = acr_select_wave<10>(pb, c, hyp_in, tid);
        c.evaluated = 1;
        if (tid == 0) { s_core = c; s_pick = pick; }
    }
    __syncthreads();
    const AcrCore s = s_core;
    const AcrPick pick = s_pick;
    if (keeper) {
        acr_keep(pb, s, pick, models_in, sorted_in, st_in, states + par, best_inliers, index_set, h_word, fin, s_full, tid, 64);
        return;
    }
    if (it >= s.cur_batch) return;                                   // the grid covers an upper bound of the batch
    const uint32_t* win = pick.best_h >= 0 ? sorted_in + (size_t)pick.best_h * pb.n : nullptr;
    const uint32_t* src = acr_sample_source(s, pick, win, best_inliers, index_set);
    uint32_t pos[5];
    clc_acr_sample_t<5>(pb.seed, (uint32_t)(s.iter + it), (uint32_t)s.n_index, pos);
    int id[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) id[j] = (int)(src ? src[pos[j]] : pos[j]);
    fpw::models_of_sample(pb.a, pb.b, pb.K1, pb.K2, id[0], id[1], id[2], id[3], id[4], pb.n, models + ((size_t)par * kSlots + (size_t)it * 10) * 18);
}

__global__ __launch_bounds__(64) void acr_solve5_kernel(AcrState* __restrict__ states /* [2] */, AcrHyp* __restrict__ hyps /* [2][slots] */,
                                                        const int par, const AcrProblem pb, uint32_t* __restrict__ sorted /* [2][slots * n] */,
                                                        double* __restrict__ models /* [2][slots * 18] */,
                                                        uint32_t* __restrict__ best_inliers, uint32_t* __restrict__ index_set,
                                                        unsigned long long* __restrict__ h_word, const AcrFinish fin)
{
    acr_solve5_body(states, hyps, par, pb, sorted, models, best_inliers, index_set, h_word, fin);
}
__global__ __launch_bounds__(64) void acr_solve5_chains_kernel(const AcrChains chains, const int par)
{
    const AcrChain& ch = chains.c[blockIdx.y];
    acr_solve5_body(ch.states, ch.hyps, par, ch.pb, ch.sorted, ch.models, ch.best_inliers, ch.index_set, ch.h_word, ch.fin);
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
    // one word per element for the exchanges that cross waves + (E <= 8) one for the exact residual bits
    const size_t lds = (T > 64 ? (size_t)P * 8 : 0) + (E <= 8 ? (size_t)P * 8 : 0);
    hipLaunchKernelGGL(acr_nfa_kernel<E>, dim3(B * pb.max_models), dim3(T), lds, stream, pb, P, d_models, d_hyp, d_sorted, d_state);
    return hipGetLastError();
}

template <int E, int KIND>
static hipError_t acr_launch_round_one(const AcrProblem& pb, int B, int P, int par, AcrState* d_states, AcrHyp* d_hyps, uint32_t* d_sorted,
                                       double* d_models, uint32_t* d_best_inliers, uint32_t* d_index_set, unsigned long long* h_word,
                                       const AcrFinish& fin, hipStream_t stream)
{
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)acr_round_kernel<E, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAcrMaxLds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const int T = P / E;
    const size_t lds = (T > 64 ? (size_t)P * 8 : 0) + (E <= 8 ? (size_t)P * 8 : 0);
    hipLaunchKernelGGL((acr_round_kernel<E, KIND>), dim3(B * AcrKind<KIND>::M + 1 /* the keeper */), dim3(T), lds, stream, d_states, d_hyps, par, P, pb,
                       d_sorted, d_models, d_best_inliers, d_index_set, h_word, fin);
    return hipGetLastError();
}
template <int KIND>
static hipError_t acr_launch_round_kind(const AcrProblem& pb, int B, int P, int par, AcrState* d_states, AcrHyp* d_hyps, uint32_t* d_sorted,
                                        double* d_models, uint32_t* d_best_inliers, uint32_t* d_index_set, unsigned long long* h_word,
                                        const AcrFinish& fin, hipStream_t stream)
{
    if (P <= 1024) return acr_launch_round_one<1, KIND>(pb, B, P, par, d_states, d_hyps, d_sorted, d_models, d_best_inliers, d_index_set, h_word, fin, stream);
    if (P == 2048) return acr_launch_round_one<2, KIND>(pb, B, P, par, d_states, d_hyps, d_sorted, d_models, d_best_inliers, d_index_set, h_word, fin, stream);
    if (P == 4096) return acr_launch_round_one<4, KIND>(pb, B, P, par, d_states, d_hyps, d_sorted, d_models, d_best_inliers, d_index_set, h_word, fin, stream);
    if (P == 8192) return acr_launch_round_one<8, KIND>(pb, B, P, par, d_states, d_hyps, d_sorted, d_models, d_best_inliers, d_index_set, h_word, fin, stream);
    return acr_launch_round_one<16, KIND>(pb, B, P, par, d_states, d_hyps, d_sorted, d_models, d_best_inliers, d_index_set, h_word, fin, stream);
}

hipError_t launch_acr_round_p3p(const AcrProblem& pb, int par, AcrState* d_states, AcrHyp* d_hyps, uint32_t* d_sorted, double* d_models,
                                uint32_t* d_best_inliers, uint32_t* d_index_set, unsigned long long* h_word, hipStream_t stream,
                                int batch_bound, uint8_t* d_mask, AcrResult* d_res, uint8_t* h_mask, int32_t* h_inliers, AcrResult* h_res)
{
    const int B = batch_bound < 1 ? 1 : (batch_bound > kAcrMaxBatch ? kAcrMaxBatch : batch_bound);
    int P = 64;
    while (P < pb.n) P <<= 1;
    const AcrFinish fin{ d_mask, d_res, h_mask, h_inliers, h_res };
    // (kind 0: P3P; 2 / 3: the seven-point / four-point models -- one launch per round for all three)
    if (pb.kind == 2) return acr_launch_round_kind<2>(pb, B, P, par, d_states, d_hyps, d_sorted, d_models, d_best_inliers, d_index_set, h_word, fin, stream);
    if (pb.kind == 3) return acr_launch_round_kind<3>(pb, B, P, par, d_states, d_hyps, d_sorted, d_models, d_best_inliers, d_index_set, h_word, fin, stream);
    if (pb.kind != 0) return hipErrorInvalidValue;
    return acr_launch_round_kind<0>(pb, B, P, par, d_states, d_hyps, d_sorted, d_models, d_best_inliers, d_index_set, h_word, fin, stream);
}

// the two-view round: acr_solve5_kernel (replay + samples + five-point; the keeper publishes the PREVIOUS round's word) and the nfa
// launch over this round's models, both on the copies of launch parity `par`
hipError_t launch_acr_round_5pt(const AcrProblem& pb, int par, AcrState* d_states, AcrHyp* d_hyps, uint32_t* d_sorted, double* d_models,
                                uint32_t* d_best_inliers, uint32_t* d_index_set, unsigned long long* h_word, hipStream_t stream,
                                int batch_bound, uint8_t* d_mask, AcrResult* d_res, uint8_t* h_mask, int32_t* h_inliers, AcrResult* h_res)
{
    const int B = batch_bound < 1 ? 1 : (batch_bound > kAcrMaxBatch ? kAcrMaxBatch : batch_bound);
    constexpr int kSlots = kAcrMaxBatch * 10;
    int P = 64;
    while (P < pb.n) P <<= 1;
    const AcrFinish fin{ d_mask, d_res, h_mask, h_inliers, h_res };
    hipLaunchKernelGGL(acr_solve5_kernel, dim3(B + 1 /* the keeper */), dim3(64), 0, stream, d_states, d_hyps, par, pb, d_sorted, d_models,
                       d_best_inliers, d_index_set, h_word, fin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const double* mo = d_models + (size_t)par * kSlots * 18;
    AcrHyp* hy = d_hyps + (size_t)par * kSlots;
    uint32_t* so = d_sorted + (size_t)par * kSlots * pb.n;
    const AcrState* stp = d_states + par;                 // what this round's keeper has just written: cur_batch = this round's batch
    if (P <= 1024) e = acr_launch_nfa<1>(pb, B, P, mo, hy, so, stp, stream);
    else if (P == 2048) e = acr_launch_nfa<2>(pb, B, P, mo, hy, so, stp, stream);
    else if (P == 4096) e = acr_launch_nfa<4>(pb, B, P, mo, hy, so, stp, stream);
    else if (P == 8192) e = acr_launch_nfa<8>(pb, B, P, mo, hy, so, stp, stream);
    else e = acr_launch_nfa<16>(pb, B, P, mo, hy, so, stp, stream);
    return e;
}

// ---- the same rounds for several solves in one launch (lockstep; pose_batch.hip drives them) ----------------------------------------------
template <typename K>
static hipError_t acr_dyn_lds(K kernel, bool (&attr_set)[64])
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAcrMaxLds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    return hipSuccess;
}
static int acr_chains_width(const AcrChains& chains, int n_chains)
{
    int P = 64;
    for (int c = 0; c < n_chains; ++c) while (P < chains.c[c].pb.n) P <<= 1;
    return P;
}
template <int E, int KIND>
static hipError_t acr_launch_round_one_chains(const AcrChains& chains, int n_chains, int B, int P, int par, hipStream_t stream)
{
    static bool attr_set[64] = {};
    const hipError_t e = acr_dyn_lds(acr_round_chains_kernel<E, KIND>, attr_set);
    if (e != hipSuccess) return e;
    const int T = P / E;
    const size_t lds = (T > 64 ? (size_t)P * 8 : 0) + (E <= 8 ? (size_t)P * 8 : 0);
    hipLaunchKernelGGL((acr_round_chains_kernel<E, KIND>), dim3(B * AcrKind<KIND>::M + 1 /* the keeper */, n_chains), dim3(T), lds, stream, chains, par, P);
    return hipGetLastError();
}
template <int KIND>
static hipError_t acr_launch_round_kind_chains(const AcrChains& chains, int n_chains, int B, int P, int par, hipStream_t stream)
{
    if (P <= 1024) return acr_launch_round_one_chains<1, KIND>(chains, n_chains, B, P, par, stream);
    if (P == 2048) return acr_launch_round_one_chains<2, KIND>(chains, n_chains, B, P, par, stream);
    if (P == 4096) return acr_launch_round_one_chains<4, KIND>(chains, n_chains, B, P, par, stream);
    if (P == 8192) return acr_launch_round_one_chains<8, KIND>(chains, n_chains, B, P, par, stream);
    return acr_launch_round_one_chains<16, KIND>(chains, n_chains, B, P, par, stream);
}
hipError_t launch_acr_round_p3p_chains(const AcrChains& chains, int n_chains, int par, int batch_bound, hipStream_t stream)
{
    if (n_chains < 1 || n_chains > kMaxBatch) return hipErrorInvalidValue;
    const int B = batch_bound < 1 ? 1 : (batch_bound > kAcrMaxBatch ? kAcrMaxBatch : batch_bound);
    const int P = acr_chains_width(chains, n_chains);                // every chain sorts at the widest chain's width: the order of its n real elements is the same
    const int kind = chains.c[0].pb.kind;                            // (a batch is one kind: check_batch / drive_group)
    for (int c = 1; c < n_chains; ++c) if (chains.c[c].pb.kind != kind) return hipErrorInvalidValue;
    if (kind == 2) return acr_launch_round_kind_chains<2>(chains, n_chains, B, P, par, stream);
    if (kind == 3) return acr_launch_round_kind_chains<3>(chains, n_chains, B, P, par, stream);
    if (kind != 0) return hipErrorInvalidValue;
    return acr_launch_round_kind_chains<0>(chains, n_chains, B, P, par, stream);
}
template <int E>
static hipError_t acr_launch_nfa_chains(const AcrChains& chains, int n_chains, int B, int P, int par, hipStream_t stream)
{
    static bool attr_set[64] = {};
    const hipError_t e = acr_dyn_lds(acr_nfa_chains_kernel<E>, attr_set);
    if (e != hipSuccess) return e;
    const int T = P / E;
    const size_t lds = (T > 64 ? (size_t)P * 8 : 0) + (E <= 8 ? (size_t)P * 8 : 0);
    hipLaunchKernelGGL(acr_nfa_chains_kernel<E>, dim3(B * 10, n_chains), dim3(T), lds, stream, chains, par, P);
    return hipGetLastError();
}
hipError_t launch_acr_round_5pt_chains(const AcrChains& chains, int n_chains, int par, int batch_bound, hipStream_t stream)
{
    if (n_chains < 1 || n_chains > kMaxBatch) return hipErrorInvalidValue;
    const int B = batch_bound < 1 ? 1 : (batch_bound > kAcrMaxBatch ? kAcrMaxBatch : batch_bound);
    const int P = acr_chains_width(chains, n_chains);
    hipLaunchKernelGGL(acr_solve5_chains_kernel, dim3(B + 1 /* the keeper */, n_chains), dim3(64), 0, stream, chains, par);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (P <= 1024) return acr_launch_nfa_chains<1>(chains, n_chains, B, P, par, stream);
    if (P == 2048) return acr_launch_nfa_chains<2>(chains, n_chains, B, P, par, stream);
    if (P == 4096) return acr_launch_nfa_chains<4>(chains, n_chains, B, P, par, stream);
    if (P == 8192) return acr_launch_nfa_chains<8>(chains, n_chains, B, P, par, stream);
    return acr_launch_nfa_chains<16>(chains, n_chains, B, P, par, stream);
}

// inputs of a solve: pinned host block -> device workspace, by a launch instead of a copy command (a copy command runs on another
// engine: 6.6 us + 8 us until the first round starts behind it, against ~3 + 3 us for a launch in the same queue)
__global__ __launch_bounds__(256) void acr_stage_kernel(const double2* __restrict__ src, double2* __restrict__ dst, const int n2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n2) dst[i] = src[i];
}
hipError_t launch_acr_stage(const double* h_pinned, double* d_dst, size_t n_doubles /* even */, hipStream_t stream)
{
    const int n2 = (int)(n_doubles / 2);
    if (n2 <= 0) return hipSuccess;
    hipLaunchKernelGGL(acr_stage_kernel, dim3((n2 + 255) / 256), dim3(256), 0, stream, (const double2*)h_pinned, (double2*)d_dst, n2);
    return hipGetLastError();
}

// the same for the solves of a lockstep batch in ONE launch (blockIdx.y = solve): eight staging launches on the shared stream were 75 us
// of a batch's 400
struct AcrStageJobs { const double2* src[kMaxBatch]; double2* dst[kMaxBatch]; int n2[kMaxBatch]; };
__global__ __launch_bounds__(256) void acr_stage_chains_kernel(const AcrStageJobs jobs)
{
    const int c = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i < jobs.n2[c]) jobs.dst[c][i] = jobs.src[c][i];
}
hipError_t launch_acr_stage_chains(const double* const* h_pinned, double* const* d_dst, const size_t* n_doubles /* even */, int n_chains,
                                   hipStream_t stream)
{
    if (n_chains < 1 || n_chains > kMaxBatch) return hipErrorInvalidValue;
    AcrStageJobs jobs{};
    int most = 0;
    for (int c = 0; c < n_chains; ++c) {
        jobs.src[c] = (const double2*)h_pinned[c]; jobs.dst[c] = (double2*)d_dst[c]; jobs.n2[c] = (int)(n_doubles[c] / 2);
        most = jobs.n2[c] > most ? jobs.n2[c] : most;
    }
    if (most <= 0) return hipSuccess;
    hipLaunchKernelGGL(acr_stage_chains_kernel, dim3((most + 255) / 256, n_chains), dim3(256), 0, stream, jobs);
    return hipGetLastError();
}

size_t acr_hyp_bytes() { return sizeof(AcrHyp); }

} // namespace clc

#if defined(CLC_ACR_STAMP)
extern "C" int clc_debug_acr_stamps(unsigned long long* out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(clc::g_acr_stamp), sizeof(unsigned long long) * 16);
}
#endif
