// k2nn.hip -- brute-force 2-NN matcher for 512-bit descriptors on gfx950 (MI355X).
//
// Semantics: reference src/CUDAK2NN.cu:46-75 (per query: running best / second-best Hamming
// distance over all train vectors in index order, accept iff second - best > threshold, ties for
// the minimum keep the lowest train index).  The ARCHITECTURE is not the reference's (a 1-D grid of
// 256 queries per block, 40 blocks at 10k queries, CUDAK2NN.cu:79, around a 32-lane shuffle butterfly).
//
// Two formulations of the same sweep live here; both produce the same {best_key, second_key} rows and share the
// in-launch finalize (atomic fold + arrival counter) and the slab/merge fallback.  Selected per context
// (CLC_K2NN_FORMULATION=popcount|matrix, default matrix); the tests run both against the oracle.
//
// (A) matrix pipe -- k2nn_sweep_mx_kernel (default).  d(q, t) = (512 - <q, t>) / 2 with bits as +-1.
//   v_mfma_scale_f32_32x32x64_f8f6f4 takes FP4 (E2M1) operands, 64 k-values per instruction, in the cycles of the bf16
//   32x32x16 form: 8 MFMAs give the 1024 distances of a 32-train x 32-query tile, 0.25 cycles per comparison per SIMD
//   against ~1.8 for the popcount form.  Everything stays exact:
//     query bit b -> nibble 0x2 | b << 3 (+1 / -1), train bit b -> nibble 0x2 | ~b << 3 (-1 / +1): sum = 2 d - 512;
//     the A (train) operand carries the E8M0 block scale 2^12 and C = 2^23 + 2^21 + (train index inside the split,
//     < 8192), so the accumulator is 2^23 + (d << 13) + index: an integer below 2^24, exact in fp32 in any order, whose
//     FLOAT BITS 0x4B000000 + (d << 13) + index are already the (distance, index) key, ordered like the floats.
//   Top-2 per query = two v_med3_f32 on the raw accumulator registers (lane = query column, the 16 accumulator
//   registers = 16 train rows); no key-forming instruction.  Descriptors stay in their 64-byte bit form in memory:
//   a wave expands its 64 queries into B operands once (registers), the workgroup expands each 32-train tile into
//   LDS (coalesced 8-byte loads, 7 VALU ops per 16 bytes of FP4) for its four waves.
//
// (B) popcount -- k2nn_sweep_kernel (round 1's kernel, kept as the measured A/B reference):
//
//   * lane = query.  Each lane keeps R complete 512-bit queries in VGPRs (16 dwords each), so a
//     distance needs NO cross-lane traffic at all.
//   * the train vector is wave-uniform: it is fetched with ONE scalar load (s_load_dwordx16, 64 B)
//     into SGPRs and used directly as the scalar operand of v_xor_b32 -- no LDS, no VGPRs, no
//     vector-memory instructions in the inner loop; the load of vector t+1 is issued before the
//     VALU work on vector t (explicit double buffer in SGPRs).  Per (query, train) pair the VALU executes
//     16 x (v_xor_b32 + v_bcnt_u32_b32 with accumulate) = 32 lane-ops.
//   * every v_xor is directly followed by the v_bcnt that consumes it (one asm statement per word).
//   * top-2 without branches or an index register: key = (distance << 22) | train_index_in_split.
//     Unsigned order on keys is (distance, index) lexicographic, so best' = min(best, key) keeps
//     the lowest index among equal distances, and second' = med3(best, second, key) is the second
//     smallest key = the second smallest distance of the multiset.  3 more lane-ops per pair.
//
// Both: 2-D decomposition (query block) x (train split) so that 10k x 10k fills 256 CUs (the reference's 1-D grid
//     would be 40 blocks).  Splits are folded into one {best_key, second_key} row per query with two
//     atomicMin (keys carry the global train index, so the fold is order-free and exact -- the
//     merge rule of SURVEY.md 8(a) note N1 with "lowest index wins" built into the key order); the
//     workgroup whose arrival completes a query block (a per-block arrival counter) applies the threshold,
//     writes the int32 results and re-arms rows and counter -- no second launch, no spinning.
//     Train sets beyond 2^22 vectors fall back to per-split slabs + an ordered merge kernel.
#include "clc_internal.h"
#include <algorithm>

namespace clc {

static constexpr int kR = 2;                 // queries per lane
static constexpr int kWaves = 8;             // waves per workgroup
static constexpr int kQPerBlock = 64 * kR;   // the kWaves waves of a workgroup share these queries and cut the train slice in kWaves
static constexpr uint32_t kKeyShift = 22;    // distance <= 512 needs 10 bits; 22 bits of index
static constexpr uint32_t kIdxMask = (1u << kKeyShift) - 1u;
static constexpr uint32_t kEmpty = 0xFFFFFFFFu;

// Pointers arrive inside a job record read from memory, so the compiler only knows them as
// generic.  Re-type them: train rows as CONSTANT address space (wave-uniform address -> s_load),
// query rows / partials as GLOBAL (global_load / global_store instead of flat_*).
// (builtin vector types: HIP's uint4 class has no constructors from qualified address spaces)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const u32x4 __attribute__((address_space(4)))* const_u4_ptr;
typedef const u32x4 __attribute__((address_space(1)))* global_cu4_ptr;
typedef u32x2 __attribute__((address_space(1)))* global_u2_ptr;

// matrix-pipe formulation
static constexpr int kMxQT = 2;                          // 32-query tiles per wave (B operands: 32 VGPRs each)
static constexpr int kMxWaves = 4;                       // waves per workgroup; they share the train tiles through LDS
static constexpr int kMxQPerBlock = 32 * kMxQT * kMxWaves;
static constexpr uint32_t kMxIdxBits = 13;               // train index inside a split rides in the accumulator: splits <= 8192 rows
static constexpr uint32_t kMxMaxSplit = 1u << kMxIdxBits;
static constexpr uint32_t kMxMagic = 0x4B000000u;        // float bits of 2^23
static constexpr uint32_t kMxInf = 0x7F800000u;          // "no key yet": above every accumulator value
static constexpr uint32_t kMxResident = 3;               // workgroups of the matrix sweep a CU holds (152 VGPRs x 4 waves, 19.5 KB of LDS)
static_assert((kK2nnXcds & (kK2nnXcds - 1u)) == 0u && kK2nnXcds == 8u, "k2nn_sweep_mx_kernel's id -> (query block, split) map is written with & 7 / >> 3");
// per-XCD unequal-share table entry (K2nnJobList.bias_tab, bias_magic == 0): query block of the XCD in bits 0..4 (at most 32 per XCD),
// first train tile in bits 5..22 (2^18 tiles = 2^23 rows: above the 2^22 rows of the atomic fold), train tiles in bits 23..31 (a
// split holds at most 256 tiles, the 13-bit index inside a split).  k2nn_plan refuses the plan when a field would not hold its value.
static constexpr uint32_t kBiasQlBits = 5, kBiasQlMask = (1u << kBiasQlBits) - 1u;
static constexpr uint32_t kBiasBeginShift = kBiasQlBits, kBiasBeginBits = 18, kBiasBeginMask = (1u << kBiasBeginBits) - 1u;
static constexpr uint32_t kBiasSizeShift = kBiasBeginShift + kBiasBeginBits, kBiasSizeMax = (1u << (32u - kBiasSizeShift)) - 1u;
static_assert(kBiasSizeMax >= kMxMaxSplit / 32u, "a split of 256 tiles must fit the size field");
static_assert(((kIdxMask + 1u) >> 5) <= kBiasBeginMask + 1u, "every tile of an atomically folded train set must fit the begin field");

int k2nn_queries_per_block(int formulation) { return formulation == K2NN_POPCOUNT ? kQPerBlock : kMxQPerBlock; }

// One train vector (16 wave-uniform dwords in SGPRs) against the R queries of this lane.
//
// Per query: 16 x (v_xor_b32 with the scalar train word, immediately consumed by v_bcnt_u32_b32 with accumulate), then
// key = (distance << 22) + index; second = med3(best, second, key); best = min(best, key).
// The exact instruction pattern is measured, not guessed (20k x 20k sweep, same device):
//     v_xor ; v_bcnt ; s_nop 0     311-313 us   <- this file
//     v_xor ; v_bcnt               368 us       (back to back: the SGPR-operand v_xor after a v_bcnt costs a full 4 cycles)
//     v_xor ; s_nop 0 ; v_bcnt     324 us       v_xor ; s_nop ; v_bcnt ; s_nop   377 us
//     ... ; s_nop 1 / two s_nop 0 / s_nop 3 after the pair   316 / 317 / 314 us
// i.e. the v_bcnt must directly follow the v_xor that feeds it and ONE idle issue slot must follow the pair: then a
// pair issues in ~6.3 cycles instead of 8 (profiles/r01_valu_issue_rates.txt).  The compiler used to supply that
// s_nop by accident (it pads every asm statement with a hazard nop); it is spelled out here so that the pattern does
// not depend on it (an s_nop inside the three top-2 instructions, or the VOP3 encoding of the v_xor, changed nothing or
// cost 1-2 %).  The first v_bcnt accumulates onto the literal 0 (no v_mov to clear the accumulator).
template <int R>
__device__ __forceinline__ void sweep_one(const uint32_t (&q)[R][16], const u32x4 a, const u32x4 b,
                                          const u32x4 c, const u32x4 d, const uint32_t t_rel,
                                          uint32_t (&best)[R], uint32_t (&second)[R])
{
#define K2NN_PAIR(t, q, accin) "v_xor_b32 %1, " t ", " q "\n\tv_bcnt_u32_b32 %0, %1, " accin "\n\ts_nop 0\n\t"
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t acc, tmp;
        asm volatile(K2NN_PAIR("%10", "%2", "0") K2NN_PAIR("%11", "%3", "%0") K2NN_PAIR("%12", "%4", "%0") K2NN_PAIR("%13", "%5", "%0")
                     K2NN_PAIR("%14", "%6", "%0") K2NN_PAIR("%15", "%7", "%0") K2NN_PAIR("%16", "%8", "%0") K2NN_PAIR("%17", "%9", "%0")
                     : "=&v"(acc), "=&v"(tmp)
                     : "v"(q[r][0]), "v"(q[r][1]), "v"(q[r][2]), "v"(q[r][3]), "v"(q[r][4]), "v"(q[r][5]), "v"(q[r][6]), "v"(q[r][7]),
                       "s"(a.x), "s"(a.y), "s"(a.z), "s"(a.w), "s"(b.x), "s"(b.y), "s"(b.z), "s"(b.w));
        asm volatile(K2NN_PAIR("%12", "%4", "%0") K2NN_PAIR("%13", "%5", "%0") K2NN_PAIR("%14", "%6", "%0") K2NN_PAIR("%15", "%7", "%0")
                     K2NN_PAIR("%16", "%8", "%0") K2NN_PAIR("%17", "%9", "%0") K2NN_PAIR("%18", "%10", "%0") K2NN_PAIR("%19", "%11", "%0")
                     "v_lshl_add_u32 %1, %0, 22, %20\n\tv_med3_u32 %3, %2, %3, %1\n\tv_min_u32 %2, %2, %1"
                     : "+v"(acc), "=&v"(tmp), "+v"(best[r]), "+v"(second[r])
                     : "v"(q[r][8]), "v"(q[r][9]), "v"(q[r][10]), "v"(q[r][11]), "v"(q[r][12]), "v"(q[r][13]), "v"(q[r][14]), "v"(q[r][15]),
                       "s"(c.x), "s"(c.y), "s"(c.z), "s"(c.w), "s"(d.x), "s"(d.y), "s"(d.z), "s"(d.w), "s"(t_rel));
    }
#undef K2NN_PAIR
}

// {best_key, second_key} of one query -> the reference's outputs (CUDAK2NN.cu:54 sentinels, :75 acceptance)
__device__ __forceinline__ void emit_result(const K2nnJobDev& job, const uint32_t qi, const uint32_t bkey, const uint32_t skey)
{
    int best_v = 100000, second_v = 200000, best_i = -1;
    if (bkey != kEmpty) {
        best_v = (int)(bkey >> kKeyShift);
        best_i = (int)(bkey & kIdxMask);
        second_v = skey == kEmpty ? 100000 : (int)(skey >> kKeyShift);
    }
    job.out[qi] = (best_i >= 0 && second_v - best_v > (int)job.thr) ? best_i : -1;
    if (job.best_out) job.best_out[qi] = (uint16_t)min(best_v, 65535);
    if (job.second_out) job.second_out[qi] = (uint16_t)min(second_v, 65535);
}

template <int R>
__global__ __launch_bounds__(64 * kWaves) void k2nn_sweep_kernel(const K2nnJobList jobs,
                                                                  uint2* __restrict__ partial)
{
    __shared__ uint32_t s_best[kWaves - 1][R][64], s_second[kWaves - 1][R][64];
    const K2nnJobDev& job = jobs.j[blockIdx.y];
    const uint32_t job_nq = k2nn_job_nq(job), job_nt = k2nn_job_nt(job);     // == job.nq / job.nt unless the counts live on the device
    // XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by linear id and every XCD has its own L2.
    // The planner makes `splits` a multiple of 8 (and with it gridDim.x), so XCD x only ever sees the train splits
    // with (split & 7) == x: each L2 holds an eighth of T plus the queries, the 8 L2s together pull 8 Q + T from HBM
    // instead of 8 (Q + T), and every XCD gets exactly the same number of workgroups.  (A 2 x 4 arrangement -- half
    // of Q, a quarter of T per XCD -- fetches less still but leaves the XCDs 8 % out of balance at 79 query blocks
    // x 78 splits: measured 2 % slower.)
    const uint32_t nblk = job.qblocks * job.splits;
    if (blockIdx.x >= nblk) return;
    const uint32_t qblock = blockIdx.x / job.splits;
    const uint32_t split = blockIdx.x - qblock * job.splits;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t qbase = qblock * (64u * R) + lane;

    uint32_t q[R][16];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t qi = qbase + 64u * r;
        if (qi >= job_nq) qi = job_nq ? job_nq - 1u : 0u;   // clamp: duplicate work, never stored (row 0 of the block always exists)
        const global_cu4_ptr qp = (global_cu4_ptr)(uintptr_t)job.q + (size_t)qi * 4u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32x4 v = qp[k];
            q[r][4 * k + 0] = v.x; q[r][4 * k + 1] = v.y; q[r][4 * k + 2] = v.z; q[r][4 * k + 3] = v.w;
        }
    }
    uint32_t best[R], second[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = kEmpty; second[r] = kEmpty; }

    // this wave's quarter [t0, t1) of the workgroup's split [s0, s1)
    const uint32_t s0 = min(split * job.t_per_split, job_nt);
    const uint32_t s1 = min(s0 + job.t_per_split, job_nt);
    const uint32_t per_wave = (s1 - s0 + kWaves - 1) / kWaves;
    const uint32_t t0 = min(s0 + wave * per_wave, s1);
    const uint32_t t1 = min(t0 + per_wave, s1);

    // Software-pipelined scalar loads: wait for the CURRENT train vector, immediately issue the
    // s_load_dwordx16 of the NEXT one, then run the 70 VALU ops of the current one while it is in
    // flight.  (Left to itself the compiler issues the load and waits for it at the loop top; the
    // explicit order is worth ~11 % at 8 waves/SIMD.)  The prefetch never reads past the slice.
    // Keys carry the index relative to s0 (the workgroup's split), so the waves' results merge by key.
    const_u4_ptr tp = (const_u4_ptr)(uintptr_t)job.t + (size_t)t0 * 4u;   // wave-uniform -> s_load_dwordx16
    if (t0 < t1) {
        // two SGPR buffers used alternately (no register copies: the scalar ALU is shared by the CU and
        // 16 s_mov per train vector made it the bottleneck); an odd tail vector is handled after the loop
        const uint32_t n = t1 - t0;
        u32x4 a0 = tp[0], b0 = tp[1], c0 = tp[2], d0 = tp[3];
        uint32_t t = t0;
        for (uint32_t pair = 0; pair < (n >> 1); ++pair, t += 2) {
            __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): buffer 0 (vector t) has landed
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 a1 = tp[4], b1 = tp[5], c1 = tp[6], d1 = tp[7];          // vector t+1 -> buffer 1
            __builtin_amdgcn_sched_barrier(0);
            sweep_one<R>(q, a0, b0, c0, d0, t - s0, best, second);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);                 // buffer 1 has landed
            __builtin_amdgcn_sched_barrier(0);
            const_u4_ptr np = (t + 2 < t1) ? tp + 8 : tp;       // vector t+2 -> buffer 0 (never past the slice)
            a0 = np[0]; b0 = np[1]; c0 = np[2]; d0 = np[3];
            __builtin_amdgcn_sched_barrier(0);
            sweep_one<R>(q, a1, b1, c1, d1, t + 1 - s0, best, second);
            __builtin_amdgcn_sched_barrier(0);
            tp += 8;
        }
        if (n & 1u) sweep_one<R>(q, a0, b0, c0, d0, t - s0, best, second);
    }

    // fold the kWaves waves through LDS: keys are unique and totally ordered, so
    // (b, s) (+) (b', s') = (min(b, b'), min(max(b, b'), s, s')) in any order
    if (wave != 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) { s_best[wave - 1][r][lane] = best[r]; s_second[wave - 1][r][lane] = second[r]; }
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < kWaves - 1; ++w) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t ob = s_best[w][r][lane], os = s_second[w][r][lane];
            second[r] = min(min(second[r], os), max(best[r], ob));
            best[r] = min(best[r], ob);
        }
    }

    if (job.atomic_merge) {
        // Fold this split into the query's global top-2 with two 32-bit atomicMin.  Keys carry the
        // GLOBAL train index here (s0 + index in split < 2^22), so unsigned order on keys is the total
        // order (distance, index) and the fold is commutative: whichever value leaves the `best` slot
        // (because a smaller key arrived) is offered to `second` by the arrival that displaced it.
        unsigned int* top = reinterpret_cast<unsigned int*>(partial + job.partial_off);
        uint32_t seen = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t qi = qbase + 64u * r;
            if (qi < job_nq && best[r] != kEmpty) {
                const uint32_t bkey = best[r] + s0;
                const uint32_t skey = second[r] == kEmpty ? kEmpty : second[r] + s0;
                const uint32_t old = atomicMin(top + 2u * qi, bkey);
                const uint32_t cand = bkey < old ? min(old, skey) : bkey;
                seen |= old;
                if (cand != kEmpty) seen |= atomicMin(top + 2u * qi + 1u, cand);
            }
        }
        // Arrival counter of this query block (armed at all-ones like the rows, so the n-th arrival reads n - 2).
        // Every atomic above RETURNS a value and the wave waits for all of them (vmcnt(0)), i.e. they have been
        // performed at the memory side -- agent-scope atomics bypass the XCD-local L2 -- before lane 0 counts this
        // workgroup in.  Whoever arrives last therefore knows every split has been folded and turns the rows into
        // results; rows and counter are read AND re-armed by atomic exchange, so no cached copy is ever consulted.
        // (A __threadfence() here would be the textbook release, but its L2 write-back + invalidate per workgroup
        // evicts the train slices the other workgroups are streaming: measured +50 % on the whole sweep.)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(seen) : : "memory");
        unsigned int* cnt = reinterpret_cast<unsigned int*>(partial) + job.cnt_off + qblock;
        uint32_t arrival = 0;
        if (lane == 0) arrival = atomicAdd(cnt, 1u) + 1u;
        arrival = __builtin_amdgcn_readfirstlane(arrival);
        if (arrival != job.splits - 1u) return;
        if (lane == 0) atomicExch(cnt, kEmpty);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t qi = qbase + 64u * r;
            if (qi < job.nq) {          // every PLANNED row gets an answer; rows past a device-side count hold no key -> -1
                const uint32_t bkey = atomicExch(top + 2u * qi, kEmpty);
                const uint32_t skey = atomicExch(top + 2u * qi + 1u, kEmpty);
                emit_result(job, qi, bkey, skey);
            }
        }
        return;
    }
    const global_u2_ptr prow = (global_u2_ptr)(uintptr_t)partial + job.partial_off + (size_t)split * job.nq_pad;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t qi = qbase + 64u * r;
        if (qi < job.nq) prow[qi] = qi < job_nq ? u32x2{ best[r], second[r] } : u32x2{ kEmpty, kEmpty };
    }
}

// ---- (A) matrix-pipe sweep ---------------------------------------------------------------------------------------
typedef int mx_v4i __attribute__((ext_vector_type(4)));
typedef int mx_v8i __attribute__((ext_vector_type(8)));
typedef float mx_v16f __attribute__((ext_vector_type(16)));
typedef const uint32_t __attribute__((address_space(1)))* global_cu32_ptr;
typedef const u32x2 __attribute__((address_space(1)))* global_cu2_ptr;

// 32 descriptor bits -> 32 FP4 nibbles 0x2 | bit << 3 (+1.0 / -1.0).  Nibble k of dword m holds bit 4 k + m: the
// k order inside a row is arbitrary as long as queries and trains use the same one.
__device__ __forceinline__ u32x4 mx_expand(const uint32_t x)
{
    u32x4 y;
    y.x = ((x << 3) & 0x88888888u) | 0x22222222u;
    y.y = ((x << 2) & 0x88888888u) | 0x22222222u;
    y.z = ((x << 1) & 0x88888888u) | 0x22222222u;
    y.w = (x & 0x88888888u) | 0x22222222u;
    return y;
}
// the same for the COMPLEMENT of x, written so that each output is one shift + one three-input bit op (v_bitop3_b32):
// complementing first and shifting the complement costs a v_not and separate and / or per output (22 instead of 14
// instructions for the 8 train bytes a thread expands per tile)
__device__ __forceinline__ u32x4 mx_expand_not(const uint32_t x)
{
    u32x4 y;
    y.x = (~(x << 3) & 0x88888888u) | 0x22222222u;
    y.y = (~(x << 2) & 0x88888888u) | 0x22222222u;
    y.z = (~(x << 1) & 0x88888888u) | 0x22222222u;
    y.w = (~x & 0x88888888u) | 0x22222222u;
    return y;
}
// Keys are positive finite floats, "none" is +inf: float order == unsigned order of the bits.  v_med3_f32 through the
// builtin, NOT inline asm: the compiler pads the MFMA-result -> VALU-read hazard only for instructions it can see
// (an asm v_med3_u32 placed straight after the last MFMA of a tile read the PREVIOUS tile's accumulator register).
__device__ __forceinline__ float mx_med3(const float a, const float b, const float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// accumulator bits -> the canonical key (distance << 22) | (t0 + index in split); kEmpty for "none" and for the
// penalised padding rows of a partial last tile (their distance field is above 512)
__device__ __forceinline__ uint32_t mx_decode(const uint32_t bits, const uint32_t t0)
{
    const uint32_t rel = bits - kMxMagic;
    const uint32_t d = rel >> kMxIdxBits;
    if (bits == kMxInf || d > 512u) return kEmpty;
    return (d << kKeyShift) | (t0 + (rel & (kMxMaxSplit - 1u)));
}

// the same for a key whose index field is relative to `end` (the row behind the split): value = 2^23 + (distance << 13) + rel,
// rel in [-8192, -1].  Through the float VALUE, not its bits: with distance 0 the value lies below 2^23, where floats step by 0.5.
__device__ __forceinline__ uint32_t mx_decode_rel(const uint32_t bits, const uint32_t end)
{
    if (bits == kMxInf) return kEmpty;
    const int v = (int)__uint_as_float(bits) - 8388608 + (int)kMxMaxSplit;          // (distance << 13) + rel + 8192, >= 0
    const uint32_t d = (uint32_t)v >> kMxIdxBits;
    if (d > 512u) return kEmpty;
    return (d << kKeyShift) | (end - kMxMaxSplit + ((uint32_t)v & (kMxMaxSplit - 1u)));
}

// STAMP: diagnostic build for the clock check (MI355X guide, DVFS item 6): lane 0 of every workgroup brackets its tile loop
// with s_memtime (shader clock) / s_memrealtime (constant 100 MHz) and stores them, with its entry / folded-in times and its
// XCC id (tools/archive/k2nn_timeline.py), in a buffer nothing else reads.  The product kernel (STAMP = false) contains no stamp.
//
// The loop (round 3): the two 32-query tiles of a wave are swept as two MFMA chains of eight, one after the other, and the
// running top-2 of a chain's accumulators is updated in the shadow of the NEXT chain's MFMAs (sched_group_barrier spells the
// interleave out: one MFMA, one A-operand read, five vector instructions).  No second accumulator set is needed -- chain 1 of tile
// t covers the top-2 of chain 0 of tile t, chain 0 of tile t + 1 covers the top-2 of chain 1 of tile t -- so the kernel keeps its
// three waves per SIMD; the price is that every A operand is read from LDS twice.  (Round 2's loop -- k-step outer, both chains per
// A read, then all 64 v_med3 -- was kept as an A/B formulation through round 5: profiles/r06_removed_variants.patch.)
// PROBE: the same code under another symbol, for the launches of the per-device share probe at context creation (capi_match.hip k2nn_probe_bias):
// a kernel trace of an application then shows ITS sweeps under k2nn_sweep_mx_kernel<false, false, false> and the probe's -- run from cold
// clocks, under four different share pairs -- under <..., true>, instead of one average over both.
template <bool STAMP, bool GLOBAL = false, bool PROBE = false>
__global__ __launch_bounds__(64 * kMxWaves) void k2nn_sweep_mx_kernel(const K2nnJobList jobs, uint2* __restrict__ partial,
                                                                      uint64_t* __restrict__ stamps)
{
    constexpr int QT = kMxQT;
    constexpr int kStride = 65;      // uint4 per k-step of a train tile in LDS: 64 lanes + 1 pad, so that the 8 k-steps
                                     // one row is expanded into land in different banks (ds_write_b128, 8 lanes per pass)
    constexpr int kQRow = 17;        // dwords per staged query row: 16 + 1 pad -> the 32 rows a half-wave reads hit 32 banks
    constexpr int kStageDw = kMxWaves * QT * 32 * kQRow;
    constexpr int kTileDw = 2 * 8 * kStride * 4;
    // one buffer, two lives: the prologue stages the raw query rows in it (per wave), the loop keeps the expanded train tiles
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kStageDw > kTileDw ? kStageDw : kTileDw];
    u32x4 (*s_a)[8 * kStride] = reinterpret_cast<u32x4 (*)[8 * kStride]>(s_mem);
    __shared__ uint32_t s_best[kMxQPerBlock], s_second[kMxQPerBlock];
    __shared__ uint32_t s_arrival;
    const K2nnJobDev& job = jobs.j[blockIdx.y];
    const uint32_t job_nq = k2nn_job_nq(job), job_nt = k2nn_job_nt(job);     // == job.nq / job.nt unless the counts live on the device
    uint64_t st_entry = 0;
    if (STAMP) st_entry = __builtin_amdgcn_s_memrealtime();
    // XCD-aware order (speed only): workgroups are dealt to the 8 XCDs round-robin by linear id, each XCD has its own
    // L2.  Workgroup L works on query block (L & 7) + 8 ((L >> 3) / splits): every split of a query block lands on ONE
    // XCD, so an XCD's L2 pulls an eighth of the queries plus the train set instead of all of both (PMC, 10k x 10k:
    // 10.1 MB fetched with the plain order = 8 x (Q + T)).  The grid is padded to a multiple of 8 query blocks.
    const uint32_t within = blockIdx.x >> 3;
    const bool biased = job.bias_a != 0u;
    // biased shares: the query blocks of an XCD are INTERLEAVED over its workgroups (query block = within % nqx), so that every query
    // block has splits on all three wave slots (blocked, as below, the 19 splits of a query block sit on one slot)
    // ... when the query blocks come in whole eights.  Otherwise (bias_magic != 0) they are interleaved over ALL workgroup ids -- query block =
    // id % qblocks, split = id / qblocks, nothing padded, a query block's splits spread over the XCDs: measured 0.8-1.5 us faster than the
    // padded per-XCD order for 36 / 34 / 28 / 47 query blocks, 0.8-1.4 us slower than the per-XCD interleave for 40 / 32 / 24 (each L2 then
    // pulls both sets whole)
    // (a template parameter, not a run-time flag: with both orders in one instantiation the register allocator copied the sixteen C registers
    // in front of every chain -- 32 v_mov per tile, +0.7 us on the 10k x 10k sweep)
    constexpr bool global_order = GLOBAL;
    const uint32_t bias_k = global_order ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__umulhi(blockIdx.x, job.bias_magic)) : 0u;   // (wave-uniform: keep it scalar)
    const uint32_t bias_q = blockIdx.x - bias_k * job.qblocks;
    const uint32_t bias_e = biased ? jobs.bias_tab[global_order ? (bias_q & 127u) : (within < 96u ? within : 95u)] : 0u;
    const uint32_t qblock = global_order ? bias_q
                          : (biased ? ((blockIdx.x + 8u - job.xcd_rot) & 7u) + 8u * (bias_e & kBiasQlMask)
                                    : ((blockIdx.x + 8u - job.xcd_rot) & 7u) + 8u * (within / job.splits));
    const uint32_t split = global_order ? bias_k : (biased ? (STAMP ? within / (job.qblocks >> 3) : 0u) : within % job.splits);
    if (qblock >= job.qblocks || (global_order && split >= job.splits) || (biased && !global_order && within >= (job.qblocks >> 3) * job.splits)) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // B operands: lane (row = l & 31, half h = l >> 5) holds the 32 k-values of words 2 j + h of its query row.
    // The wave's 64 rows are fetched with COALESCED 16-byte loads (4 KB = four wave instructions) and handed round
    // through the wave's own LDS staging area; every lane then picks its 8 words per tile.  (Each lane loading its own
    // words straight from memory -- 16 strided dword loads touching 32 cache lines per instruction -- took 11 k cycles
    // per wave, a quarter of the kernel, measured with in-kernel stamps: tools/archive/k2nn_mfma.hip.)
    mx_v4i b[QT][8];
    {
        uint32_t* stage = s_mem + wave * (QT * 32 * kQRow);
        const global_cu4_ptr qbase = (global_cu4_ptr)(uintptr_t)job.q;
#pragma unroll
        for (int i = 0; i < 2 * QT; ++i) {
            const uint32_t idx = (uint32_t)i * 64u + lane;               // 16-byte chunk of the wave's QT x 32 rows
            const uint32_t r_local = idx >> 2, chunk = idx & 3u;
            uint32_t row = qblock * kMxQPerBlock + wave * (QT * 32u) + r_local;
            if (row >= job_nq) row = job_nq ? job_nq - 1u : 0u;          // clamp: duplicate work, never stored (row 0 of the block always exists)
            const u32x4 v = qbase[(size_t)row * 4u + chunk];
            uint32_t* d = stage + r_local * kQRow + chunk * 4u;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const uint32_t* src = stage + ((uint32_t)qt * 32u + (lane & 31u)) * kQRow + (lane >> 5);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x4 v = mx_expand(src[2 * j]);
                b[qt][j] = mx_v4i{ (int)v.x, (int)v.y, (int)v.z, (int)v.w };
            }
        }
    }
    __syncthreads();                 // the staging area becomes the train-tile buffer
    // C of the first MFMA of a tile: 2^23 + 2^21 + index (inside the TILE) of the lane's 16 train rows; the same for every tile
    // (C/D layout of the 32x32 forms: column = lane & 31, row = 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3))
    float cinit[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[i] = 8388608.0f + 2097152.0f + (float)(8 * (i >> 2) + 4 * (int)(lane >> 5) + (i & 3));
    float best[QT], second[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { best[qt] = __uint_as_float(kMxInf); second[qt] = __uint_as_float(kMxInf); }

    // this workgroup's train rows [s0, s1): t_per_split is a multiple of 32, so only the train set's last tile can be partial
    uint32_t s0 = min(split * job.t_per_split, job_nt);
    uint32_t s1 = min(s0 + job.t_per_split, job_nt);
    if (biased && !global_order) {
        s0 = min(((bias_e >> kBiasBeginShift) & kBiasBeginMask) * 32u, job_nt);
        s1 = min(s0 + (bias_e >> kBiasSizeShift) * 32u, job_nt);
    }
    if (global_order) {
        // the query block's splits in id order: n0 on slot 0 with bias_a tiles each, n1 on slot 1 with bias_b, the rest on slot 2 with c
        // (the first `extra` of them c + 1) tiles; the last split ends with the train set
        const uint32_t n0 = bias_e & 0xFFu, n1 = (bias_e >> 8) & 0xFFu, c = (bias_e >> 16) & 0xFFu, extra = bias_e >> 24;
        const uint32_t k0 = min(split, n0), k1 = min(split - k0, n1), k2 = split - k0 - k1;
        const uint32_t begin = k0 * job.bias_a + k1 * job.bias_b + k2 * c + min(k2, extra);
        const uint32_t size = split < n0 ? job.bias_a : (split < n0 + n1 ? job.bias_b : c + (k2 < extra ? 1u : 0u));
        s0 = min(begin * 32u, job_nt);
        s1 = split + 1u == job.splits ? job_nt : min(s0 + size * 32u, job_nt);
    }
    if (global_order) {
        s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s0);
        s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s1);
    }
    const uint32_t ntiles = (s1 - s0 + 31u) >> 5;
    const int scale_a = 0x8B8B8B8B, scale_b = 0x7F7F7F7F;            // E8M0 block scales: 2^12 (trains), 1 (queries)
    // train tile = 32 rows x 64 B = 2 KB contiguous: thread tid owns its 8 bytes number tid = row tid >> 3, words
    // 2 j', 2 j' + 1 (j' = tid & 7), i.e. both lane halves of k-step j' of that row
    const uint32_t my_row = tid >> 3, my_j = tid & 7u;
    const uint32_t dst = my_j * kStride + my_row;
    const global_cu2_ptr tbase = (global_cu2_ptr)(uintptr_t)job.t + (size_t)my_j;
    const uint32_t last_row = job_nt ? job_nt - 1u : 0u;             // in a register: read through `job` it is a scalar load
                                                                     // (+ wait) from the kernel arguments in every tile
    auto load_bits = [&](const uint32_t tile) -> u32x2 {
        const uint32_t row = min(s0 + tile * 32u + my_row, last_row);   // stays in bounds; such rows are penalised through C
        return tbase[(size_t)row * 8u];
    };
    u32x2 r0 = u32x2{ 0u, 0u }, r1 = u32x2{ 0u, 0u };
    if (ntiles) { r0 = load_bits(0u); r1 = load_bits(min(1u, ntiles - 1u)); }   // (an empty split -- device-side count -- reads nothing)
    mx_v16f acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = __uint_as_float(kMxInf); acc1[i] = __uint_as_float(kMxInf); }
    uint64_t st_clk = 0, st_real = 0;
    if (STAMP) { st_clk = __builtin_amdgcn_s_memtime(); st_real = __builtin_amdgcn_s_memrealtime(); }
    for (uint32_t t = 0; t < ntiles; ++t) {
        const uint32_t buf = t & 1u;
        s_a[buf][dst] = mx_expand_not(r0.x);
        s_a[buf][dst + 32] = mx_expand_not(r0.y);
        r0 = r1;
        __syncthreads();          // one barrier per tile: buffer `buf` was last read two iterations ago, before the previous barrier
        if (t + 2u < ntiles) r1 = load_bits(t + 2u);                // bits of the tile after next: in flight for a whole tile
        if (t + 1u == ntiles && ((s1 - s0) & 31u)) {
            // partial last tile: rows past the end get a penalty that puts their distance field above 512
            const uint32_t valid = (s1 - s0) & 31u;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((uint32_t)(8 * (i >> 2) + 4 * (int)(lane >> 5) + (i & 3)) >= valid) cinit[i] += 4194304.0f + 8192.0f;
        }
        static_assert(QT == 2, "the pipelined loop is written for two chains");
        // chain c of this tile (8 MFMAs) with the top-2 update of `prev` (the accumulators of the chain before it) in its shadow
        // sixteen steps: step s = MFMA j = s & 7 of chain s >> 3; the A operand of step s + kAhead is requested at step s (an LDS
        // read takes longer than one MFMA: with the next step's operand only, every MFMA waited for its read)
#ifndef CLC_K2NN_AHEAD
#define CLC_K2NN_AHEAD 3
#endif
        constexpr int kAhead = CLC_K2NN_AHEAD;
        u32x4 ring[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) ring[k] = s_a[buf][k * kStride + lane];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int c = st >> 3, j = st & 7;
            const u32x4 av = ring[st % kAhead];
            const mx_v8i a8 = { (int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0 };
            const mx_v8i b8 = { b[c][j].x, b[c][j].y, b[c][j].z, b[c][j].w, 0, 0, 0, 0 };
            if (st + kAhead < 16) ring[st % kAhead] = s_a[buf][((st + kAhead) & 7) * kStride + lane];
            mx_v16f& acc = c ? acc1 : acc0;
            const mx_v16f& prev = c ? acc0 : acc1;       // chain 0 folds in chain 1 of the tile before, chain 1 folds in chain 0
            const int pq = c ? 0 : 1;
            mx_v16f ci;
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) ci[i] = cinit[i];
            } else ci = acc;
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, ci, 4, 4, 0, scale_a, 0, scale_b);
            // two of the sixteen keys of the chain before, in this MFMA's shadow (32 cycles of matrix pipe, 8 of them holding the
            // vector issue: four v_med3 fit)
#pragma unroll
            for (int i = 2 * j; i < 2 * j + 2; ++i) {
                second[pq] = mx_med3(best[pq], second[pq], prev[i]);
                best[pq] = mx_med3(best[pq], prev[i], 0.0f);
            }
            // the step back of the chain just folded in: instead of advancing the index part of the sixteen C registers per tile, the
            // running pair steps BACK by one tile -- its index field is then relative to the tile about to be processed (negative for
            // the tiles behind; the integer stays exact in fp32, order and tie rule unchanged: a lower global index is a lower value)
            if (j == 7) { best[pq] -= 32.0f; second[pq] -= 32.0f; }
            __builtin_amdgcn_sched_barrier(0);                           // keep the groups in this order
        }
    }
    {                  // the last tile's second chain is still to be folded in
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            second[1] = mx_med3(best[1], second[1], acc1[i]);
            best[1] = mx_med3(best[1], acc1[i], 0.0f);
        }
        best[1] -= 32.0f; second[1] -= 32.0f;
    }

    if (STAMP) {
        const uint64_t en_clk = __builtin_amdgcn_s_memtime(), en_real = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            uint64_t* o = stamps + 8u * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = st_clk; o[1] = st_real; o[2] = en_clk; o[3] = en_real; o[4] = st_entry;
        }
    }
    // lanes l and l ^ 32 hold the same query column (different train rows): fold them, decode to canonical keys with the
    // index relative to s0 (slab mode) or global (atomic mode), and hand the workgroup's 256 results over through LDS so
    // that thread i finishes query i of the block with coalesced rows
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const uint32_t mb = __float_as_uint(best[qt]), ms = __float_as_uint(second[qt]);
        const uint32_t ob = __shfl_xor(mb, 32), os = __shfl_xor(ms, 32);
        if (lane < 32u) {
            const uint32_t slot = (wave * QT + qt) * 32u + lane;
            // after the last tile's step back every index is relative to the row BEHIND the split: -(rows of the split) .. -1
            const uint32_t end = (job.atomic_merge ? s0 : 0u) + 32u * ntiles;
            s_best[slot] = mx_decode_rel(min(mb, ob), end);
            s_second[slot] = mx_decode_rel(min(min(ms, os), max(mb, ob)), end);
        }
    }
    __syncthreads();
    const uint32_t qi = qblock * kMxQPerBlock + tid;
    const uint32_t bkey_l = s_best[tid], skey_l = s_second[tid];
    if (!job.atomic_merge) {
        const global_u2_ptr prow = (global_u2_ptr)(uintptr_t)partial + job.partial_off + (size_t)split * job.nq_pad;
        if (qi < job.nq) prow[qi] = qi < job_nq ? u32x2{ bkey_l, skey_l } : u32x2{ kEmpty, kEmpty };
        return;
    }
    // same fold / arrival / finalize protocol as the popcount kernel (see there for why no fence is needed): every
    // thread's atomics RETURN, every wave waits for them, the workgroup barrier joins the four waves, then one lane
    // counts the workgroup in; the last arrival of a query block turns the rows into results and re-arms them.
    unsigned int* top = reinterpret_cast<unsigned int*>(partial + job.partial_off);
    uint32_t seen = 0;
    if (qi < job_nq && bkey_l != kEmpty) {
        const uint32_t old = atomicMin(top + 2u * qi, bkey_l);
        const uint32_t cand = bkey_l < old ? min(old, skey_l) : bkey_l;
        seen |= old;
        if (cand != kEmpty) seen |= atomicMin(top + 2u * qi + 1u, cand);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(seen) : : "memory");
    __syncthreads();
    unsigned int* cnt = reinterpret_cast<unsigned int*>(partial) + job.cnt_off + qblock;
    if (tid == 0) s_arrival = atomicAdd(cnt, 1u) + 1u;
    __syncthreads();
    if (STAMP && tid == 0) {          // folded in: everything but the finalize of the last arrival
        uint64_t* o = stamps + 8u * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = ((uint64_t)qblock << 32) | split;
        // XCC_ID[3:0] | HW_ID << 8 (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]: where on the chip this workgroup's wave 0 sits)
        o[7] = (uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11)) |
               ((uint64_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11)) << 8);
    }
    if (s_arrival != job.splits - 1u) return;
    if (tid == 0) atomicExch(cnt, kEmpty);
    if (qi < job.nq) {
        const uint32_t bkey = atomicExch(top + 2u * qi, kEmpty);
        const uint32_t skey = atomicExch(top + 2u * qi + 1u, kEmpty);
        emit_result(job, qi, bkey, skey);
    }
}

// Atomic mode, jobs with an EMPTY train set only (no sweep workgroup exists that could finalize them):
// the reference leaves best_i uninitialised there (CUDAK2NN.cu:54,75); defined here as "no match".
__global__ __launch_bounds__(256) void k2nn_nomatch_kernel(const K2nnJobList jobs)
{
    const K2nnJobDev& job = jobs.j[blockIdx.y];
    const uint32_t qi = blockIdx.x * 256u + threadIdx.x;
    if (job.nt != 0u || qi >= job.nq) return;
    emit_result(job, qi, kEmpty, kEmpty);
}

// Fold the per-split partials of one query and threshold.
// The merge of two partial results is associative as long as the left operand holds the lower
// train indices, so the splits of a query are folded by kMergeGroups threads in parallel (each a
// contiguous run of splits, loads issued 4 deep) and the group results are folded in order through
// LDS.  Reads stay coalesced: consecutive lanes own consecutive queries of one split row.
static constexpr int kMergeQ = 32;        // queries per workgroup
static constexpr int kMergeGroups = 8;    // split groups per query

struct Top2 { int best_v, second_v, best_i; };

__device__ __forceinline__ void fold(Top2& a, const uint32_t bk, const uint32_t sk, const uint32_t t0)
{
    if (bk == kEmpty) return;
    const int b_best = (int)(bk >> kKeyShift);
    const int b_idx = (int)(t0 + (bk & kIdxMask));
    const int b_second = sk == kEmpty ? 100000 : (int)(sk >> kKeyShift);
    if (b_best < a.best_v) {
        a.second_v = min(a.best_v, b_second);
        a.best_v = b_best;
        a.best_i = b_idx;
    } else {
        a.second_v = min(a.second_v, b_best);
    }
}

__global__ __launch_bounds__(kMergeQ * kMergeGroups) void k2nn_merge_kernel(const K2nnJobList jobs,
                                                                             const uint2* __restrict__ partial)
{
    __shared__ Top2 sh[kMergeGroups][kMergeQ];
    const K2nnJobDev& job = jobs.j[blockIdx.y];
    const uint32_t qx = threadIdx.x & (kMergeQ - 1);
    const uint32_t g = threadIdx.x / kMergeQ;
    const uint32_t qi = blockIdx.x * kMergeQ + qx;
    if (blockIdx.x * kMergeQ >= job.nq) return;                       // whole workgroup past the end
    const uint32_t nsplit = job.nt ? job.splits : 0u;                  // nt == 0: nothing was swept -> -1
    const uint32_t per_g = (nsplit + kMergeGroups - 1) / kMergeGroups;
    const uint32_t s0 = min(g * per_g, nsplit), s1 = min(s0 + per_g, nsplit);
    Top2 a{ 100000, 200000, -1 };                                      // CUDAK2NN.cu:54 sentinels
    if (qi < job.nq) {
        const uint2* p = partial + job.partial_off + (size_t)s0 * job.nq_pad + qi;
        uint32_t s = s0;
        for (; s + 4 <= s1; s += 4, p += 4 * (size_t)job.nq_pad) {
            const uint2 e0 = p[0], e1 = p[job.nq_pad], e2 = p[2 * (size_t)job.nq_pad], e3 = p[3 * (size_t)job.nq_pad];
            fold(a, e0.x, e0.y, s * job.t_per_split);
            fold(a, e1.x, e1.y, (s + 1) * job.t_per_split);
            fold(a, e2.x, e2.y, (s + 2) * job.t_per_split);
            fold(a, e3.x, e3.y, (s + 3) * job.t_per_split);
        }
        for (; s < s1; ++s, p += job.nq_pad) {
            const uint2 e = *p;
            fold(a, e.x, e.y, s * job.t_per_split);
        }
    }
    sh[g][qx] = a;
    __syncthreads();
    if (g != 0 || qi >= job.nq) return;
    for (int k = 1; k < kMergeGroups; ++k) {                           // in order: group k holds higher indices
        const Top2 b = sh[k][qx];
        if (b.best_i < 0) continue;
        if (b.best_v < a.best_v) {
            a.second_v = min(a.best_v, b.second_v);
            a.best_v = b.best_v;
            a.best_i = b.best_i;
        } else {
            a.second_v = min(a.second_v, b.best_v);
        }
    }
    job.out[qi] = (a.best_i >= 0 && a.second_v - a.best_v > (int)job.thr) ? a.best_i : -1;
    if (job.best_out) job.best_out[qi] = (uint16_t)min(a.best_v, 65535);
    if (job.second_out) job.second_out[qi] = (uint16_t)min(a.second_v, 65535);
}

K2nnPlan k2nn_plan(K2nnJobDev* jobs, int njobs, int target_blocks, bool xcd_map, int formulation, int bias_a, int bias_b, K2nnDevice dev)
{
    // Everything below that speaks of XCDs, CUs or wave slots takes the numbers from the DEVICE (hipDeviceAttributeNumberOfXccs,
    // multiProcessorCount; clc_ctx_create) -- rounds 3-4 had 8 / 32 / 768 written in.  The sweep kernel's workgroup -> (query block, split)
    // map is compiled for kK2nnXcds XCDs; it is a permutation of the work on any device, but the balance arguments (one XCD's share of the
    // slots, the slot a workgroup id lands on) only hold where the device has that many: elsewhere the plan keeps equal shares and skips
    // the per-XCD fit.
    const bool xcd_known = dev.n_xcd == kK2nnXcds && dev.n_cu % kK2nnXcds == 0u;
    const uint32_t cu_x = dev.n_cu / kK2nnXcds;                    // CUs of one XCD = workgroups per wave slot and XCD (32 on MI355X)
    const uint32_t slots = kMxResident * dev.n_cu;                // resident workgroups of the matrix sweep (768)

    const bool mx = formulation != K2NN_POPCOUNT;
    const uint32_t qpb = mx ? kMxQPerBlock : kQPerBlock;
    K2nnPlan plan{0, true};
    for (int j = 0; j < njobs; ++j)
        if (jobs[j].nt > kIdxMask + 1u) plan.atomic_merge = false;
    uint32_t total_qblocks = 0;
    for (int j = 0; j < njobs; ++j) {
        jobs[j].qblocks = (jobs[j].nq + qpb - 1) / qpb;
        total_qblocks += jobs[j].qblocks;
    }
    if (total_qblocks == 0) total_qblocks = 1;
    // popcount: about target_blocks workgroups, several rounds of them; matrix: at most target_blocks (= the resident
    // slots, 3 workgroups per CU), so that the whole grid runs as ONE round -- a partial second round costs a full
    // round's time at a fraction of the machine
    uint32_t want = mx ? (uint32_t)target_blocks / total_qblocks : ((uint32_t)target_blocks + total_qblocks - 1) / total_qblocks;
    if (mx) {
        // Every split of a query block runs on ONE XCD (the kernel's XCD-aware order), so the round must also fit per XCD: an eighth
        // of the slots against the most query blocks any XCD takes.  34 query blocks (8.5 k queries) at 768 / 34 = 22 splits put
        // 5 x 22 = 110 workgroups on the 96 slots of XCDs 0 and 1 -- a second round there, measured 33 us instead of 26 for such a
        // grid.  Jobs continue round the XCDs where the previous one stopped (xcd_rot), so the maximum is ceil(total / 8).
        const uint32_t per_xcd = (total_qblocks + kK2nnXcds - 1u) / kK2nnXcds;
        const uint32_t fit = (uint32_t)target_blocks / kK2nnXcds / per_xcd;
        if (xcd_known && fit < want) want = fit;
    }
    if (want < 1) want = 1;
    size_t off = 0;
    uint32_t rot = 0;
    for (int j = 0; j < njobs; ++j) {
        K2nnJobDev& jb = jobs[j];
        jb.xcd_rot = mx ? rot : 0u;
        rot = (rot + jb.qblocks) & (kK2nnXcds - 1u);
        uint32_t splits = want;
        // popcount: >= 16 train vectors per wave; matrix: >= 2 tiles of 32 per workgroup
        const uint32_t min_per = mx ? 64u : 16u * kWaves;
        const uint32_t max_splits = jb.nt / min_per > 0 ? jb.nt / min_per : 1u;
        if (splits > max_splits) splits = max_splits;
        // popcount: a multiple of 8 splits pins train split s to XCD s & 7 (see that kernel).  The matrix kernel reads the
        // 64-byte rows themselves (a 10k-row set is 640 KB: every XCD's L2 holds all of it), so it keeps the exact count.
        if (!mx && xcd_map && splits >= 8u) splits = (splits + 4u) / 8u * 8u > max_splits ? splits / 8u * 8u : (splits + 4u) / 8u * 8u;
        uint32_t per = jb.nt ? (jb.nt + splits - 1) / splits : 1u;
        if (mx) {
            per = (per + 31u) & ~31u;                                     // whole tiles of 32 train rows
            if (per > kMxMaxSplit) per = kMxMaxSplit;                     // 13-bit index inside a split (in the accumulator)
        }
        if (per > kIdxMask + 1u) per = kIdxMask + 1u;                     // index field is 22 bits
        splits = jb.nt ? (jb.nt + per - 1) / per : 1u;
        jb.splits = splits;
        jb.t_per_split = per;
        jb.bias_a = jb.bias_b = jb.bias_magic = 0u;
        jb.slot_wgs = 0u;
        jb.nq_pad = (jb.nq + 63u) & ~63u;
        jb.partial_off = (uint32_t)off;
        jb.atomic_merge = plan.atomic_merge ? 1u : 0u;
        off += plan.atomic_merge ? (size_t)jb.nq_pad : (size_t)splits * jb.nq_pad;
        if (plan.atomic_merge) {                                          // arrival counters: one uint32 per query block
            jb.cnt_off = (uint32_t)(2u * off);
            off += (jb.qblocks + 1u) / 2u;
        }
    }
    // (one job only: the slot a workgroup id lands on is a property of the FIRST resident round of a launch -- the workgroups of a
    // launch's later jobs are dispatched as slots fall free, in no fixed relation to their ids; multi-pair launches also amortise the
    // tail the unequal shares remove: 22 us per pair in a 6-pair launch against 25 for one pair alone)
    if (mx && njobs == 1 && plan.atomic_merge && bias_a > 0 && bias_b > 0 && xcd_known && kMxResident * cu_x <= 96u) {
        // unequal shares by wave slot (K2nnJobDev.bias_a): one job whose grid is one round of three workgroups per CU -- every XCD holds
        // more than 64 and at most 96 of them -- with whole eights of query blocks (nothing padded) and counts known on the host
        K2nnJobDev& jb = jobs[0];
        const uint32_t per_xcd = (jb.qblocks >> 3) * jb.splits;
        const uint32_t nt_tiles = (jb.nt + 31u) >> 5;
        const uint32_t mean = jb.splits ? nt_tiles / jb.splits : 0u;
        // (with one or two query blocks per XCD the biased plan measured 0.3-0.8 us SLOWER -- 2048 x 60000, 4096 x 30000 --, from three on 1-2 us faster)
        if ((jb.qblocks & 7u) == 0u && jb.qblocks >= 24u && per_xcd > 2u * cu_x && per_xcd <= 3u * cu_x && !jb.cnt_q && !jb.cnt_t && jb.splits >= 6u && mean >= 8u) {
            // bias_a / bias_b arrive in 1/256 of the equal share (values below 64: tiles, for experiments)
            const uint32_t den = jb.splits * 256u;
            const uint32_t a = bias_a < 64 ? (uint32_t)bias_a : (nt_tiles * (uint32_t)bias_a + den / 2u) / den;
            const uint32_t b = bias_b < 64 ? (uint32_t)bias_b : (nt_tiles * (uint32_t)bias_b + den / 2u) / den;
            // every query block's split sizes as the kernel derives them: all within the 13-bit index field (256 tiles), the slot-0 and
            // slot-1 splits inside the train set
            const uint32_t nqx = jb.qblocks >> 3;
            // ... and everything the 32-bit table entry carries inside its field (a 12-bit begin field silently clamped train
            // sets beyond 4096 tiles), no slot-2 split below the two tiles the equal plan guarantees (min_per)
            bool ok = a >= 1u && b >= 1u && a <= 255u && b <= 255u && nqx <= kBiasQlMask + 1u && nt_tiles <= kBiasBeginMask;
            for (uint32_t ql = 0; ok && ql < nqx; ++ql) {
                const uint32_t n0 = std::min((cu_x - ql + nqx - 1u) / nqx, jb.splits);
                const uint32_t n1 = std::min((2u * cu_x - ql + nqx - 1u) / nqx, jb.splits) - n0;
                const uint32_t n2 = jb.splits - n0 - n1;
                const uint32_t used = n0 * a + n1 * b;
                ok = n2 >= 1u && used < nt_tiles && (nt_tiles - used + n2 - 1u) / n2 <= 255u && (nt_tiles - used) / n2 >= 2u;
            }
            if (ok) { jb.bias_a = a; jb.bias_b = b; jb.bias_magic = 0u; jb.slot_wgs = cu_x; }
        }
        // the same for query-block counts that are NOT whole eights: interleaved over all workgroup ids (K2nnJobDev.bias_magic), the split
        // count taken anew (no per-XCD fit, nothing padded)
        const uint32_t QB = jb.qblocks;
        uint32_t sp = QB ? (uint32_t)target_blocks / QB : 0u;
        if (sp > nt_tiles / 8u) sp = nt_tiles / 8u;                            // >= 8 tiles per split on average
        if (jb.bias_a == 0u && (QB & 7u) != 0u && QB >= 24u && QB <= 128u && sp >= 6u && QB * sp > 2u * dev.n_cu && QB * sp <= slots && !jb.cnt_q && !jb.cnt_t &&
            (uint32_t)target_blocks == slots) {
            const uint32_t den = sp * 256u;
            const uint32_t a = bias_a < 64 ? (uint32_t)bias_a : (nt_tiles * (uint32_t)bias_a + den / 2u) / den;
            const uint32_t b = bias_b < 64 ? (uint32_t)bias_b : (nt_tiles * (uint32_t)bias_b + den / 2u) / den;
            bool ok = a >= 1u && b >= 1u && a <= 255u && b <= 255u;
            for (uint32_t q = 0; ok && q < QB; ++q) {
                const uint32_t n0 = std::min((dev.n_cu - q + QB - 1u) / QB, sp);
                const uint32_t n1 = std::min((2u * dev.n_cu - q + QB - 1u) / QB, sp) - n0;
                const uint32_t n2 = sp - n0 - n1;
                const uint32_t used = n0 * a + n1 * b;
                ok = n2 >= 1u && used < nt_tiles && (nt_tiles - used + n2 - 1u) / n2 <= 255u && (nt_tiles - used) / n2 >= 2u;
            }
            if (ok) {
                jb.bias_a = a; jb.bias_b = b;
                jb.bias_magic = (uint32_t)((0x100000000ull + QB - 1u) / QB);
                jb.slot_wgs = dev.n_cu;
                jb.splits = sp;
                jb.t_per_split = ((nt_tiles + sp - 1u) / sp) * 32u;             // (the equal share: reporting only)
            }
        }
    }
    plan.partial_elems = off;
    return plan;
}

hipError_t launch_k2nn(const K2nnJobDev* jobs, int njobs, uint2* d_partial, hipStream_t stream, Profiler* prof, int formulation,
                       uint64_t* d_stamps, bool probe)
{
    for (int base = 0; base < njobs; base += kK2nnJobsPerLaunch) {
        const int cnt = njobs - base < kK2nnJobsPerLaunch ? njobs - base : kK2nnJobsPerLaunch;
        K2nnJobList list;
        uint32_t grid_x = 0, max_nq = 0, max_nq_empty = 0;
        for (int j = 0; j < cnt; ++j) {
            list.j[j] = jobs[base + j];
            // matrix kernel: query blocks are dealt to XCDs in eights (see the kernel) -> pad the count to a multiple of 8
            const uint32_t qb = formulation == K2NN_POPCOUNT ? list.j[j].qblocks : ((list.j[j].qblocks + 7u) & ~7u);
            const uint32_t gx = list.j[j].nt ? qb * list.j[j].splits : 0u;
            if (gx > grid_x) grid_x = gx;
            if (list.j[j].nq > max_nq) max_nq = list.j[j].nq;
            if (list.j[j].nt == 0u && list.j[j].nq > max_nq_empty) max_nq_empty = list.j[j].nq;
        }
        for (int j = cnt; j < kK2nnJobsPerLaunch; ++j) list.j[j] = K2nnJobDev{};
        for (uint32_t& e : list.bias_tab) e = 0u;
        if (cnt == 1 && list.j[0].bias_a != 0u && list.j[0].bias_magic != 0u && formulation != K2NN_POPCOUNT) {
            // interleaved over all ids: per query block q its split k is workgroup k QB + q, on wave slot (k QB + q) / 256 -- n0 splits on slot 0
            // (bits 0..7), n1 on slot 1 (8..15), tiles of a slot-2 split (16..23), how many of them take one more (24..31)
            const K2nnJobDev& jb = list.j[0];
            const uint32_t QB = jb.qblocks, nt_tiles = (jb.nt + 31u) >> 5;
            for (uint32_t q = 0; q < QB && q < 128u; ++q) {
                const uint32_t n0 = std::min((jb.slot_wgs - q + QB - 1u) / QB, jb.splits);
                const uint32_t n1 = std::min((2u * jb.slot_wgs - q + QB - 1u) / QB, jb.splits) - n0;
                const uint32_t n2 = jb.splits - n0 - n1;
                const uint32_t used = n0 * jb.bias_a + n1 * jb.bias_b;
                const uint32_t rest = nt_tiles > used ? nt_tiles - used : 0u;
                const uint32_t c = n2 ? rest / n2 : 0u, extra = n2 ? rest - c * n2 : 0u;
                list.bias_tab[q] = n0 | (n1 << 8) | (std::min(c, 255u) << 16) | (std::min(extra, 255u) << 24);
            }
            grid_x = QB * jb.splits;                                          // nothing padded
        } else
        if (cnt == 1 && list.j[0].bias_a != 0u && formulation != K2NN_POPCOUNT) {
            // workgroup w of an XCD: query block w % nqx (the query blocks of an XCD interleaved, so that each has splits on all three wave
            // slots), its split k = w / nqx sits on slot w / 32: n0 splits of bias_a tiles, n1 of bias_b, the n2 slot-2 splits share the
            // rest (the first `extra` of them one tile more)
            const K2nnJobDev& jb = list.j[0];
            const uint32_t nqx = jb.qblocks >> 3, nt_tiles = (jb.nt + 31u) >> 5;
            for (uint32_t w = 0; w < nqx * jb.splits && w < 96u; ++w) {
                const uint32_t ql = w % nqx, k = w / nqx;
                const uint32_t n0 = std::min((jb.slot_wgs - ql + nqx - 1u) / nqx, jb.splits);
                const uint32_t n1 = std::min((2u * jb.slot_wgs - ql + nqx - 1u) / nqx, jb.splits) - n0;
                const uint32_t n2 = jb.splits - n0 - n1;
                const uint32_t used = n0 * jb.bias_a + n1 * jb.bias_b;
                const uint32_t rest = nt_tiles > used ? nt_tiles - used : 0u;
                const uint32_t c = n2 ? rest / n2 : 0u, extra = n2 ? rest - c * n2 : 0u;
                const uint32_t k0 = std::min(k, n0), k1 = std::min(k - k0, n1), k2 = k - k0 - k1;
                const uint32_t begin = k0 * jb.bias_a + k1 * jb.bias_b + k2 * c + std::min(k2, extra);
                uint32_t size = k < n0 ? jb.bias_a : (k < n0 + n1 ? jb.bias_b : c + (k2 < extra ? 1u : 0u));
                if (k + 1u == jb.splits) size = nt_tiles > begin ? nt_tiles - begin : 0u;      // the last split takes what is left
                // (k2nn_plan has checked that every field holds its value; the clamps only keep a corrupted job record from aliasing)
                list.bias_tab[w] = (ql & kBiasQlMask) | (std::min(begin, kBiasBeginMask) << kBiasBeginShift) | (std::min(size, kBiasSizeMax) << kBiasSizeShift);
            }
        }
        if (max_nq == 0) continue;
        if (grid_x > 0) {
            prof_mark(prof, CLC_KERNEL_K2NN_SWEEP, true, stream);
            if (formulation == K2NN_POPCOUNT)
                hipLaunchKernelGGL(k2nn_sweep_kernel<kR>, dim3(grid_x, cnt), dim3(64 * kWaves), 0, stream, list, d_partial);
            else if (cnt == 1 && list.j[0].bias_a != 0u && list.j[0].bias_magic != 0u && d_stamps)
                hipLaunchKernelGGL((k2nn_sweep_mx_kernel<true, true>), dim3(grid_x, cnt), dim3(64 * kMxWaves), 0, stream, list, d_partial, d_stamps);
            else if (cnt == 1 && list.j[0].bias_a != 0u && list.j[0].bias_magic != 0u)
                hipLaunchKernelGGL((k2nn_sweep_mx_kernel<false, true>), dim3(grid_x, cnt), dim3(64 * kMxWaves), 0, stream, list, d_partial,
                                   (uint64_t*)nullptr);
            else if (d_stamps)
                hipLaunchKernelGGL((k2nn_sweep_mx_kernel<true>), dim3(grid_x, cnt), dim3(64 * kMxWaves), 0, stream, list, d_partial, d_stamps);
            else if (probe)
                hipLaunchKernelGGL((k2nn_sweep_mx_kernel<false, false, true>), dim3(grid_x, cnt), dim3(64 * kMxWaves), 0, stream, list, d_partial,
                                   (uint64_t*)nullptr);
            else
                hipLaunchKernelGGL((k2nn_sweep_mx_kernel<false>), dim3(grid_x, cnt), dim3(64 * kMxWaves), 0, stream, list, d_partial,
                                   (uint64_t*)nullptr);
            prof_mark(prof, CLC_KERNEL_K2NN_SWEEP, false, stream);
        }
        if (list.j[0].atomic_merge) {
            if (max_nq_empty)
                hipLaunchKernelGGL(k2nn_nomatch_kernel, dim3((max_nq_empty + 255) / 256, cnt), dim3(256), 0, stream, list);
        } else {
            prof_mark(prof, CLC_KERNEL_K2NN_MERGE, true, stream);
            hipLaunchKernelGGL(k2nn_merge_kernel, dim3((max_nq + kMergeQ - 1) / kMergeQ, cnt), dim3(kMergeQ * kMergeGroups),
                               0, stream, list, (const uint2*)d_partial);
            prof_mark(prof, CLC_KERNEL_K2NN_MERGE, false, stream);
        }
    }
    return hipGetLastError();
}

} // namespace clc
