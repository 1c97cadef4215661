// k2nn_mfma.hip -- experiment: the K2NN all-pairs sweep on the matrix pipe, bit-identical to the
// popcount formulation.  Standalone harness: expands descriptors, sweeps, checks every (best index, best, second)
// against a CPU brute force, times the launches.  Not shipped; the winner moves into coloc_amd/csrc/k2nn.hip.
//
// Formulation.  Hamming distance of two 512-bit rows = (512 - <q, t>) / 2 with bits mapped to +-1.  gfx950's
// v_mfma_scale_f32_32x32x64_f8f6f4 takes FP4 (E2M1) operands, 64 k-values per instruction, at the bf16 32x32x16
// cycle count (32 cycles per SIMD): 8 MFMAs turn a 32-train x 32-query tile of bit rows into 1024 exact distances.
//   query bit b -> nibble 0x2 | b << 3   (+1 / -1)
//   train bit b -> nibble 0x2 | ~b << 3  (-1 / +1)      => sum of products = 2 d - 512
//   A (trains) carries the E8M0 block scale 2^12, so the accumulator is C + 4096 (2 d - 512) = C - 2^21 + (d << 13).
//   C = 2^23 + 2^21 + (train index inside the split, < 8192): the accumulator's FLOAT BITS are then
//   0x4B000000 + (d << 13) + index -- already the (distance, index) key, ordered like an unsigned integer.
//   Every partial sum is an integer below 2^24, so fp32 accumulation is exact in any order.
// Top-2 per query = v_med3_u32 + v_min_u32 on the raw accumulator registers (lane = query column, the 16 accumulator
// registers = 16 train rows), no key-forming instruction at all.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/k2mfma tools/k2nn_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

static constexpr uint32_t kMagic = 0x4B000000u;          // bits of 2^23
static constexpr int kIdxBits = 13;                      // train index inside a split
static constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;
static constexpr uint32_t kInfBits = 0x7F800000u;

// ---- expansion: n rows of 16 dwords -> tiles of 32 rows in MFMA operand order -------------------------------------
// out layout (uint4 units): [tile][kstep j: 8][lane: 64], lane = h * 32 + row, holding the 32 nibbles of word 2 j + h.
__global__ __launch_bounds__(256) void expand_kernel(const uint32_t* __restrict__ desc, uint32_t n, uint32_t n_tiles,
                                                     u32x4* __restrict__ out, uint32_t flip)
{
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= n_tiles * 512u) return;
    const uint32_t lane = gid & 63u, j = (gid >> 6) & 7u, tile = gid >> 9;
    const uint32_t row = tile * 32u + (lane & 31u), w = 2u * j + (lane >> 5);
    u32x4 y = { 0u, 0u, 0u, 0u };                         // fp4 zeros for the padding rows
    if (row < n) {
        const uint32_t x = desc[(size_t)row * 16u + w] ^ flip;
        y.x = ((x << 3) & 0x88888888u) | 0x22222222u;
        y.y = ((x << 2) & 0x88888888u) | 0x22222222u;
        y.z = ((x << 1) & 0x88888888u) | 0x22222222u;
        y.w = (x & 0x88888888u) | 0x22222222u;
    }
    out[gid] = y;
}

// Keys are positive finite floats (bits 0x4B000000 + ...), "empty" is +inf: float order == unsigned order of the bits.
// v_med3_f32 through the builtin, NOT inline asm: the compiler pads the MFMA-result -> VALU-read hazard only for
// instructions it knows (an asm v_med3_u32 straight after the last MFMA read the previous tile's accumulator).
__device__ __forceinline__ float fmed3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// ---- sweep --------------------------------------------------------------------------------------------------------
// workgroup = 4 waves; wave w owns query tiles (qblock * 4 + w) * QT .. + QT - 1; all waves share the train tiles of
// the split through LDS (double-buffered, one barrier per tile).
template <int QT>
__global__ __launch_bounds__(256) void sweep_kernel(const u32x4* __restrict__ Qx, const u32x4* __restrict__ Tx, uint32_t nq,
                                                    uint32_t nt, uint32_t q_tiles, uint32_t splits, uint32_t tiles_per_split,
                                                    uint2* __restrict__ partial, uint32_t nq_pad)
{
    __shared__ u32x4 s_a[2][512];
    const uint32_t qblock = blockIdx.x / splits, split = blockIdx.x - qblock * splits;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t t_tiles = (nt + 31u) >> 5;
    const uint32_t tb = split * tiles_per_split, te = min(tb + tiles_per_split, t_tiles);

    v4i b[QT][8];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        uint32_t tile = (qblock * 4u + wave) * QT + qt;
        if (tile >= q_tiles) tile = q_tiles - 1u;          // duplicate work, never stored
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4 v = Qx[(size_t)tile * 512u + j * 64u + lane];
            b[qt][j] = v4i{ (int)v.x, (int)v.y, (int)v.z, (int)v.w };
        }
    }
    // C of the first MFMA of each tile: 2^23 + 2^21 + index of this lane's 16 train rows inside the split
    float cinit[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[i] = 8388608.0f + 2097152.0f + (float)(8 * (i >> 2) + 4 * (int)(lane >> 5) + (i & 3));
    float best[QT], second[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { best[qt] = __uint_as_float(kInfBits); second[qt] = __uint_as_float(kInfBits); }

    const int scale_a = 0x8B8B8B8B, scale_b = 0x7F7F7F7F;   // E8M0: 2^12 and 1
    u32x4 r0 = { 0u, 0u, 0u, 0u }, r1 = { 0u, 0u, 0u, 0u };
    if (tb < te) { r0 = Tx[(size_t)tb * 512u + tid]; r1 = Tx[(size_t)tb * 512u + 256u + tid]; }
    for (uint32_t t = tb; t < te; ++t) {
        const uint32_t buf = (t - tb) & 1u;
        s_a[buf][tid] = r0;
        s_a[buf][tid + 256u] = r1;
        __syncthreads();
        if (t + 1u < te) { r0 = Tx[(size_t)(t + 1u) * 512u + tid]; r1 = Tx[(size_t)(t + 1u) * 512u + 256u + tid]; }
        if (t + 1u == t_tiles && (nt & 31u)) {
            // last tile of the train set is partial: rows past the end get a penalty above every real distance
            const uint32_t valid = nt & 31u;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((uint32_t)(8 * (i >> 2) + 4 * (int)(lane >> 5) + (i & 3)) >= valid) cinit[i] += 4194304.0f + 8192.0f;
        }
        v16f acc[QT];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4 av = s_a[buf][j * 64 + lane];
            const v4i a = { (int)av.x, (int)av.y, (int)av.z, (int)av.w };
            const v8i a8 = { a.x, a.y, a.z, a.w, 0, 0, 0, 0 };
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const v8i b8 = { b[qt][j].x, b[qt][j].y, b[qt][j].z, b[qt][j].w, 0, 0, 0, 0 };
                v16f c;
                if (j == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) c[i] = cinit[i];
                } else c = acc[qt];
                acc[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, scale_a, 0, scale_b);
            }
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                second[qt] = fmed3(best[qt], second[qt], acc[qt][i]);
                best[qt] = fmed3(best[qt], acc[qt][i], 0.0f);          // = min: every key is > 0
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) cinit[i] += 32.0f;
    }

    // lanes l and l ^ 32 hold the same query column (different train rows): fold, then decode
    const uint32_t t0 = tb * 32u;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const uint32_t mb = __float_as_uint(best[qt]), ms = __float_as_uint(second[qt]);
        const uint32_t ob = __shfl_xor(mb, 32), os = __shfl_xor(ms, 32);
        const uint32_t s = min(min(ms, os), max(mb, ob));
        const uint32_t bb = min(mb, ob);
        const uint32_t q = ((qblock * 4u + wave) * QT + qt) * 32u + (lane & 31u);
        if (lane < 32u && q < nq) {
            auto decode = [&](uint32_t key) -> uint32_t {
                const uint32_t rel = key - kMagic;
                const uint32_t d = rel >> kIdxBits;
                if (key == kInfBits || d > 512u) return kEmptyKey;
                return (d << 22) | (t0 + (rel & ((1u << kIdxBits) - 1u)));
            };
            partial[(size_t)split * nq_pad + q] = make_uint2(decode(bb), decode(s));
        }
    }
}

// ---- fused sweep: descriptors stay in their 64-byte bit form in memory; the workgroup expands them itself -----------
// * prologue: lane (row = l & 31, h = l >> 5) of a wave loads words 2 j + h of its query rows and expands them into
//   the B operands (registers, loop invariant);
// * per train tile: thread tid loads the uint2 number tid of the tile's 2 KB (coalesced; row tid >> 3, words 2 j', 2 j' + 1,
//   j' = tid & 7 = exactly the two lane halves of k-step j'), expands it into two uint4 and stores them at
//   [j'][h][row]; the k-step stride is 65 uint4 so that the eight j' of a row land in different banks.
//   The raw bits of the tile two iterations ahead are in flight in registers (2 VGPRs per tile).
__device__ __forceinline__ u32x4 expand32(uint32_t x)
{
    u32x4 y;
    y.x = ((x << 3) & 0x88888888u) | 0x22222222u;
    y.y = ((x << 2) & 0x88888888u) | 0x22222222u;
    y.z = ((x << 1) & 0x88888888u) | 0x22222222u;
    y.w = (x & 0x88888888u) | 0x22222222u;
    return y;
}

template <int QT, int XCD, int ABL = 0>
__global__ __launch_bounds__(256) void fused_kernel(const uint32_t* __restrict__ Q, const uint32_t* __restrict__ T, uint32_t nq,
                                                    uint32_t nt, uint32_t qblocks, uint32_t splits, uint32_t tiles_per_split,
                                                    uint2* __restrict__ partial, uint32_t nq_pad)
{
    constexpr int kStride = 65;                              // uint4 per k-step in LDS (64 + 1 pad)
    __shared__ u32x4 s_a[2][8 * kStride];
    const uint64_t tk_entry = ABL == 9 ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t rt_entry = ABL == 9 ? __builtin_amdgcn_s_memrealtime() : 0;
    uint32_t bid = blockIdx.x;
    if (XCD) {                                               // consecutive ids round-robin over 8 XCDs: give XCD x the splits = x mod 8
        const uint32_t per = gridDim.x >> 3;                 // host makes gridDim.x a multiple of 8 and splits a multiple of 8
        bid = (bid & 7u) + ((bid >> 3) << 3);                // identity; kept for clarity: bid % 8 == split % 8 because splits % 8 == 0
        (void)per;
    }
    const uint32_t qblock = bid / splits, split = bid - qblock * splits;
    if (qblock >= qblocks) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t t_tiles = (nt + 31u) >> 5;
    const uint32_t tb = split * tiles_per_split, te = min(tb + tiles_per_split, t_tiles);
    if (tb >= te) return;

    v4i b[QT][8];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        uint32_t row = ((qblock * 4u + wave) * QT + qt) * 32u + (lane & 31u);
        if (row >= nq) row = nq - 1u;                        // duplicate work, never stored
        const uint32_t* qp = Q + (size_t)row * 16u + (lane >> 5);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4 v = expand32(qp[2 * j]);
            b[qt][j] = v4i{ (int)v.x, (int)v.y, (int)v.z, (int)v.w };
        }
    }
    float cinit[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[i] = 8388608.0f + 2097152.0f + (float)(8 * (i >> 2) + 4 * (int)(lane >> 5) + (i & 3));
    float best[QT], second[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { best[qt] = __uint_as_float(kInfBits); second[qt] = __uint_as_float(kInfBits); }

    const int scale_a = 0x8B8B8B8B, scale_b = 0x7F7F7F7F;   // E8M0: 2^12 and 1
    const uint32_t my_row = tid >> 3, my_j = tid & 7u;
    const uint32_t dst = my_j * kStride + my_row;            // h = 0 slot; h = 1 is 32 further
    auto load_bits = [&](uint32_t t) -> uint2 {
        uint32_t row = t * 32u + my_row;
        if (row >= nt) row = nt - 1u;                         // stays in bounds; masked below by the penalty in C
        return *reinterpret_cast<const uint2*>(T + (size_t)row * 16u + 2u * my_j);
    };
    uint2 r0 = load_bits(tb), r1 = load_bits(min(tb + 1u, te - 1u));
    uint64_t ph[5] = { 0, 0, 0, 0, 0 }, tk0 = 0, tk_start = 0;
    if (ABL == 9) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tk_start = __builtin_amdgcn_s_memtime(); }
#define STAMP(i) if (ABL == 9) { const uint64_t now = __builtin_amdgcn_s_memtime(); ph[i] += now - tk0; tk0 = now; }
    for (uint32_t t = tb; t < te; ++t) {
        if (ABL == 9) tk0 = __builtin_amdgcn_s_memtime();
        const uint32_t buf = (t - tb) & 1u;
        if (ABL != 3) {
            s_a[buf][dst] = expand32(~r0.x);
            s_a[buf][dst + 32] = expand32(~r0.y);
        }
        r0 = r1;
        if (ABL == 9) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        STAMP(0)
        if (ABL != 3 && ABL != 5) __syncthreads();
        STAMP(1)
        if (ABL != 4 && t + 2u < te) r1 = load_bits(t + 2u);
        if (t + 1u == t_tiles && (nt & 31u)) {
            const uint32_t valid = nt & 31u;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((uint32_t)(8 * (i >> 2) + 4 * (int)(lane >> 5) + (i & 3)) >= valid) cinit[i] += 4194304.0f + 8192.0f;
        }
        v16f acc[QT];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4 av = ABL == 3 ? u32x4{ r0.x + j, r0.y, r1.x, r1.y } : s_a[buf][j * kStride + lane];
            const v8i a8 = { (int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0 };
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const v8i b8 = { b[qt][j].x, b[qt][j].y, b[qt][j].z, b[qt][j].w, 0, 0, 0, 0 };
                v16f c;
                if (j == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) c[i] = cinit[i];
                } else c = acc[qt];
                if (ABL == 2) { acc[qt] = c; acc[qt][j] += __int_as_float(a8[0] ^ b8[1]); }
                else acc[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, scale_a, 0, scale_b);
            }
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                second[qt] = fmed3(best[qt], second[qt], acc[qt][i]);
                best[qt] = fmed3(best[qt], acc[qt][i], 0.0f);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) cinit[i] += 32.0f;
        if (ABL == 9) { asm volatile("" : "+v"(best[0]), "+v"(second[0]), "+v"(best[QT - 1]), "+v"(second[QT - 1])); }
        STAMP(3)
    }
    const uint64_t tk_loop_end = ABL == 9 ? __builtin_amdgcn_s_memtime() : 0;

    const uint32_t t0 = tb * 32u;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const uint32_t mb = __float_as_uint(best[qt]), ms = __float_as_uint(second[qt]);
        const uint32_t ob = __shfl_xor(mb, 32), os = __shfl_xor(ms, 32);
        const uint32_t s = min(min(ms, os), max(mb, ob));
        const uint32_t bb = min(mb, ob);
        const uint32_t q = ((qblock * 4u + wave) * QT + qt) * 32u + (lane & 31u);
        if (lane < 32u && q < nq) {
            auto decode = [&](uint32_t key) -> uint32_t {
                const uint32_t rel = key - kMagic;
                const uint32_t d = rel >> kIdxBits;
                if (key == kInfBits || d > 512u) return kEmptyKey;
                return (d << 22) | (t0 + (rel & ((1u << kIdxBits) - 1u)));
            };
            partial[(size_t)split * nq_pad + q] = make_uint2(decode(bb), decode(s));
        }
    }
    if (ABL == 9) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            uint64_t* o = reinterpret_cast<uint64_t*>(partial) + (size_t)nq_pad * splits + ((size_t)blockIdx.x * 4 + wave) * 8;
            o[0] = ph[0]; o[1] = ph[1]; o[2] = tk_start - tk_entry; o[3] = ph[3]; o[4] = tk_loop_end - tk_start; o[5] = te - tb;
            o[6] = __builtin_amdgcn_s_memtime() - tk_loop_end; o[7] = rt_entry;
        }
    }
}

// ---- pipelined fused sweep: the top-2 of tile t runs in the shadow of tile t + 1's MFMAs ---------------------------
// Two accumulator sets per query tile; the loop body is unrolled by two so that the sets alternate without copies.
// An MFMA blocks the SIMD's vector issue for 8 of its 32 cycles: 4 v_med3 + ~2 other VALU fit in the rest, so ONE wave
// per SIMD can in principle keep the matrix pipe busy.  sched_group_barrier spells the interleave out for the scheduler.
template <int QT>
__global__ __launch_bounds__(256) void pipe_kernel(const uint32_t* __restrict__ Q, const uint32_t* __restrict__ T, uint32_t nq,
                                                   uint32_t nt, uint32_t qblocks, uint32_t splits, uint32_t tiles_per_split,
                                                   uint2* __restrict__ partial, uint32_t nq_pad)
{
    constexpr int kStride = 65;
    __shared__ u32x4 s_a[2][8 * kStride];
    const uint32_t bid = blockIdx.x;
    const uint32_t qblock = bid / splits, split = bid - qblock * splits;
    if (qblock >= qblocks) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t t_tiles = (nt + 31u) >> 5;
    const uint32_t tb = split * tiles_per_split, te = min(tb + tiles_per_split, t_tiles);
    if (tb >= te) return;
    const uint32_t ntiles = te - tb;

    v4i b[QT][8];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        uint32_t row = ((qblock * 4u + wave) * QT + qt) * 32u + (lane & 31u);
        if (row >= nq) row = nq - 1u;
        const uint32_t* qp = Q + (size_t)row * 16u + (lane >> 5);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4 v = expand32(qp[2 * j]);
            b[qt][j] = v4i{ (int)v.x, (int)v.y, (int)v.z, (int)v.w };
        }
    }
    v16f cinit;
#pragma unroll
    for (int i = 0; i < 16; ++i) cinit[i] = 8388608.0f + 2097152.0f + (float)(8 * (i >> 2) + 4 * (int)(lane >> 5) + (i & 3));
    float best[QT], second[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { best[qt] = __uint_as_float(kInfBits); second[qt] = __uint_as_float(kInfBits); }
    const int scale_a = 0x8B8B8B8B, scale_b = 0x7F7F7F7F;
    const uint32_t my_row = tid >> 3, my_j = tid & 7u;
    const uint32_t dst = my_j * kStride + my_row;
    auto load_bits = [&](uint32_t t) -> uint2 {
        uint32_t row = (tb + t) * 32u + my_row;
        if (row >= nt) row = nt - 1u;
        return *reinterpret_cast<const uint2*>(T + (size_t)row * 16u + 2u * my_j);
    };
    uint2 r0 = load_bits(0u), r1 = load_bits(min(1u, ntiles - 1u));
    auto stage = [&](uint32_t t) {                      // expand tile t (bits in r0) into LDS, queue the bits of t + 2
        const uint32_t buf = t & 1u;
        s_a[buf][dst] = expand32(~r0.x);
        s_a[buf][dst + 32] = expand32(~r0.y);
        r0 = r1;
        __syncthreads();
        if (t + 2u < ntiles) r1 = load_bits(t + 2u);
        if (tb + t + 1u == t_tiles && (nt & 31u)) {
            const uint32_t valid = nt & 31u;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((uint32_t)(8 * (i >> 2) + 4 * (int)(lane >> 5) + (i & 3)) >= valid) cinit[i] += 4194304.0f + 8192.0f;
        }
    };
    // MFMAs of tile t into `cur`, top-2 of the previous tile's `old` interleaved (have_old = false: prologue)
    auto compute = [&](uint32_t t, v16f (&cur)[QT], v16f (&old)[QT], const bool have_old) {
        const uint32_t buf = t & 1u;
        v4i a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const u32x4 av = s_a[buf][j * kStride + lane];
            a[j] = v4i{ (int)av.x, (int)av.y, (int)av.z, (int)av.w };
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const v8i a8 = { a[j].x, a[j].y, a[j].z, a[j].w, 0, 0, 0, 0 };
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const v8i b8 = { b[qt][j].x, b[qt][j].y, b[qt][j].z, b[qt][j].w, 0, 0, 0, 0 };
                cur[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, j == 0 ? cinit : cur[qt], 4, 4, 0, scale_a, 0, scale_b);
            }
        }
        if (have_old) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    second[qt] = fmed3(best[qt], second[qt], old[qt][i]);
                    best[qt] = fmed3(best[qt], old[qt][i], 0.0f);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) cinit[i] += 32.0f;
        // the interleave: per MFMA four top-2 instructions of the previous tile (+ the odd cinit add)
        if (have_old) {
#pragma unroll
            for (int k = 0; k < 8 * QT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (32 * QT + 16) / (8 * QT), 0);
            }
        }
    };
    v16f accA[QT], accB[QT];
    stage(0u);
    compute(0u, accA, accB, false);
    uint32_t t = 1;
    for (; t + 1u < ntiles; t += 2) {
        stage(t);
        compute(t, accB, accA, true);
        stage(t + 1u);
        compute(t + 1u, accA, accB, true);
    }
    v16f (*last)[QT] = &accA;
    if (t < ntiles) { stage(t); compute(t, accB, accA, true); last = &accB; }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            second[qt] = fmed3(best[qt], second[qt], (*last)[qt][i]);
            best[qt] = fmed3(best[qt], (*last)[qt][i], 0.0f);
        }
    }

    const uint32_t t0 = tb * 32u;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const uint32_t mb = __float_as_uint(best[qt]), ms = __float_as_uint(second[qt]);
        const uint32_t ob = __shfl_xor(mb, 32), os = __shfl_xor(ms, 32);
        const uint32_t s = min(min(ms, os), max(mb, ob));
        const uint32_t bb = min(mb, ob);
        const uint32_t q = ((qblock * 4u + wave) * QT + qt) * 32u + (lane & 31u);
        if (lane < 32u && q < nq) {
            auto decode = [&](uint32_t key) -> uint32_t {
                const uint32_t rel = key - kMagic;
                const uint32_t d = rel >> kIdxBits;
                if (key == kInfBits || d > 512u) return kEmptyKey;
                return (d << 22) | (t0 + (rel & ((1u << kIdxBits) - 1u)));
            };
            partial[(size_t)split * nq_pad + q] = make_uint2(decode(bb), decode(s));
        }
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------
static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static void cpu_top2(const std::vector<uint32_t>& Q, const std::vector<uint32_t>& T, int nq, int nt, std::vector<int>& bi,
                     std::vector<int>& bv, std::vector<int>& sv)
{
    bi.assign(nq, -1); bv.assign(nq, 100000); sv.assign(nq, 200000);
#pragma omp parallel for
    for (int q = 0; q < nq; ++q) {
        int best_v = 100000, second_v = 200000, best_i = -1;
        for (int t = 0; t < nt; ++t) {
            int d = 0;
            for (int k = 0; k < 16; ++k) d += __builtin_popcount(Q[(size_t)q * 16 + k] ^ T[(size_t)t * 16 + k]);
            second_v = std::min(d, second_v);
            if (d < best_v) { second_v = best_v; best_i = t; best_v = d; }
        }
        bi[q] = best_i; bv[q] = best_v; sv[q] = second_v;
    }
}

template <int QT, int MODE>
static int run_case(int nq, int nt, int target_blocks, int reps, bool verify)
{
    std::vector<uint32_t> Q((size_t)nq * 16), T((size_t)nt * 16);
    for (auto& v : T) v = (uint32_t)rnd();
    for (auto& v : Q) v = (uint32_t)rnd();
    // planted near-duplicates, exact duplicates (ties), complements (d = 512)
    for (int q = 0; q < nq; q += 3) {
        const int t = (int)(rnd() % (uint64_t)nt);
        memcpy(&Q[(size_t)q * 16], &T[(size_t)t * 16], 64);
        const int flips = (int)(rnd() % 61);
        for (int f = 0; f < flips; ++f) { const int bit = (int)(rnd() % 512); Q[(size_t)q * 16 + bit / 32] ^= 1u << (bit % 32); }
    }
    if (nt > 8) { memcpy(&T[(size_t)5 * 16], &T[(size_t)2 * 16], 64); memcpy(&T[(size_t)(nt - 1) * 16], &T[(size_t)2 * 16], 64); }
    if (nq > 4) for (int k = 0; k < 16; ++k) Q[(size_t)4 * 16 + k] = ~T[(size_t)(nt / 2) * 16 + k];

    const uint32_t q_tiles = (nq + 31) / 32, t_tiles = (nt + 31) / 32;
    const uint32_t qblocks = (q_tiles + 4 * QT - 1) / (4 * QT);
    uint32_t splits = std::max(1u, (uint32_t)target_blocks / qblocks);
    splits = std::min(splits, t_tiles);
    uint32_t tps = (t_tiles + splits - 1) / splits;
    tps = std::min(tps, 256u);                                  // 13-bit index inside a split
    splits = (t_tiles + tps - 1) / tps;
    if (MODE == 2 && splits >= 8) { splits = splits / 8 * 8; tps = (t_tiles + splits - 1) / splits; }
    const uint32_t nq_pad = (nq + 63) & ~63;

    uint32_t *dQ, *dT; u32x4 *dQx, *dTx; uint2* dP;
    CHECK(hipMalloc(&dQ, Q.size() * 4)); CHECK(hipMalloc(&dT, T.size() * 4));
    CHECK(hipMalloc(&dQx, (size_t)q_tiles * 8192)); CHECK(hipMalloc(&dTx, (size_t)t_tiles * 8192));
    const size_t stamp_words = (size_t)qblocks * splits * 4 * 8;
    CHECK(hipMalloc(&dP, (size_t)splits * nq_pad * 8 + stamp_words * 8));
    CHECK(hipMemcpy(dQ, Q.data(), Q.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dT, T.data(), T.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(dP, 0xFF, (size_t)splits * nq_pad * 8));

    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    float ms_expand = 0.f, ms_sweep = 0.f;
    for (int rep = 0; rep < reps + 2; ++rep) {
        CHECK(hipEventRecord(e0));
        if (MODE == 0) {
            hipLaunchKernelGGL(expand_kernel, dim3((q_tiles * 512 + 255) / 256), dim3(256), 0, 0, dQ, (uint32_t)nq, q_tiles, dQx, 0u);
            hipLaunchKernelGGL(expand_kernel, dim3((t_tiles * 512 + 255) / 256), dim3(256), 0, 0, dT, (uint32_t)nt, t_tiles, dTx, 0xFFFFFFFFu);
        }
        CHECK(hipEventRecord(e1));
        if (MODE == 0)
            hipLaunchKernelGGL(sweep_kernel<QT>, dim3(qblocks * splits), dim3(256), 0, 0, dQx, dTx, (uint32_t)nq, (uint32_t)nt, q_tiles,
                               splits, tps, dP, nq_pad);
        else if (MODE == 3)
            hipLaunchKernelGGL((pipe_kernel<QT>), dim3(qblocks * splits), dim3(256), 0, 0, dQ, dT, (uint32_t)nq, (uint32_t)nt,
                               qblocks, splits, tps, dP, nq_pad);
        else if (MODE >= 10)
            hipLaunchKernelGGL((fused_kernel<QT, 0, MODE - 10>), dim3(qblocks * splits), dim3(256), 0, 0, dQ, dT, (uint32_t)nq, (uint32_t)nt,
                               qblocks, splits, tps, dP, nq_pad);
        else
            hipLaunchKernelGGL((fused_kernel<QT, MODE == 2>), dim3(qblocks * splits), dim3(256), 0, 0, dQ, dT, (uint32_t)nq, (uint32_t)nt,
                               qblocks, splits, tps, dP, nq_pad);
        CHECK(hipEventRecord(e2));
        CHECK(hipEventSynchronize(e2));
        float a, b;
        CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
        if (rep >= 2) { ms_expand += a; ms_sweep += b; }
    }
    CHECK(hipGetLastError());
    ms_expand /= reps; ms_sweep /= reps;
    printf("mode=%d QT=%d nq=%d nt=%d qblocks=%u splits=%u tiles/split=%u : expand %.2f us, sweep %.2f us, %.1f Gcmp/s\n", MODE, QT, nq, nt, qblocks,
           splits, tps, ms_expand * 1e3, ms_sweep * 1e3, (double)nq * nt / (ms_sweep * 1e-3) * 1e-9);

    if (MODE == 19) {
        std::vector<uint64_t> st(stamp_words);
        CHECK(hipMemcpy(st.data(), (char*)dP + (size_t)splits * nq_pad * 8, stamp_words * 8, hipMemcpyDeviceToHost));
        double sum[8] = {};
        size_t nw = 0;
        uint64_t rt_min = ~0ull, rt_max = 0;
        for (size_t w = 0; w < (size_t)qblocks * splits * 4; ++w) {
            if (st[w * 8 + 5] == 0) continue;
            const double nt_ = (double)st[w * 8 + 5];
            sum[0] += st[w * 8 + 0] / nt_; sum[1] += st[w * 8 + 1] / nt_; sum[3] += st[w * 8 + 3] / nt_;
            sum[2] += (double)st[w * 8 + 2]; sum[4] += (double)st[w * 8 + 4]; sum[6] += (double)st[w * 8 + 6];
            sum[5] += nt_;
            rt_min = std::min(rt_min, st[w * 8 + 7]); rt_max = std::max(rt_max, st[w * 8 + 7]);
            ++nw;
        }
        printf("  stamps, mean over %zu waves: per tile: expand+ds_write %.0f | barrier %.0f | rest (loads issue, ds_read, MFMA, top-2) %.0f cycles ; per wave: prologue %.0f, loop %.0f (%.1f tiles), epilogue %.0f cycles ; wave start spread %.2f us\n",
               nw, sum[0] / nw, sum[1] / nw, sum[3] / nw, sum[2] / nw, sum[4] / nw, sum[5] / nw, sum[6] / nw, (double)(rt_max - rt_min) / 100.0);
    }
    int bad = 0;
    if (verify) {
        std::vector<uint2> P((size_t)splits * nq_pad);
        CHECK(hipMemcpy(P.data(), dP, P.size() * 8, hipMemcpyDeviceToHost));
        std::vector<int> bi, bv, sv;
        cpu_top2(Q, T, nq, nt, bi, bv, sv);
        for (int q = 0; q < nq; ++q) {
            uint32_t b = kEmptyKey, s = kEmptyKey;
            for (uint32_t sp = 0; sp < splits; ++sp) {
                const uint2 p = P[(size_t)sp * nq_pad + q];
                s = std::min(std::min(s, p.y), std::max(b, p.x));
                b = std::min(b, p.x);
            }
            const int gbi = b == kEmptyKey ? -1 : (int)(b & 0x3FFFFF), gbv = b == kEmptyKey ? 100000 : (int)(b >> 22);
            const int gsv = s == kEmptyKey ? (b == kEmptyKey ? 200000 : 100000) : (int)(s >> 22);
            const int esv = sv[q] >= 100000 ? (bi[q] < 0 ? 200000 : 100000) : sv[q];
            if (gbi != bi[q] || gbv != bv[q] || gsv != esv) {
                if (bad < 10) printf("  MISMATCH q=%d gpu (i=%d b=%d s=%d) cpu (i=%d b=%d s=%d)\n", q, gbi, gbv, gsv, bi[q], bv[q], esv);
                ++bad;
            }
        }
        printf("  verify: %d mismatches of %d queries\n", bad, nq);
    }
    CHECK(hipFree(dQ)); CHECK(hipFree(dT)); CHECK(hipFree(dQx)); CHECK(hipFree(dTx)); CHECK(hipFree(dP));
    return bad;
}

int main(int argc, char** argv)
{
    const int target = argc > 1 ? atoi(argv[1]) : 768;
    int bad = 0;
    bad += run_case<2, 1>(10000, 10000, target, 20, true);
    for (int rep = 0; rep < 2; ++rep) {
        run_case<2, 1>(10000, 10000, target, 20, false);
        run_case<2, 19>(10000, 10000, target, 20, false);
        run_case<1, 19>(10000, 10000, target, 20, false);
        run_case<2, 19>(10000, 10000, 256, 20, false);
    }
    printf(bad ? "FAILED: %d mismatches\n" : "ALL OK\n", bad);
    return bad ? 1 : 0;
}
