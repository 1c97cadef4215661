// clatch.hip -- 512-bit LATCH binary descriptor on gfx950 (MI355X).
//
// Semantics: reference src/CLATCH.cu:157-188 with the learned arrangement of
// include/coloc/CLATCH.h:170 (re-encoded in latch_pattern.inc):
//   * rotated 64x64 window around the keypoint, point-sampled from the keypoint's pyramid level
//     with clamp addressing: sx = (int)((pt.x + (xo*c - yo*s)) + 0.5f), sy likewise (:166), fp32 in
//     source order, no FMA contraction (file built with -ffp-contract=off);
//   * 512 tests S_n = sum over an 8x8 patch of (A-B)^2 - (C-B)^2 (int32 exact), bit n = (S_n < 0),
//     bit n&31 of little-endian uint32 word n>>5 (:185-188).
//   * (s, c) = clc_sincosf(angle): the correctly rounded fp32 sine / cosine (clc_sincos.h).
//
// The ARCHITECTURE is not the reference's (512-thread block per keypoint, 32-lane butterflies):
//   * one WAVE64 per keypoint and one wave per workgroup: no barrier anywhere, 12.7 KB of LDS per
//     wave, so 12 keypoints are in flight per CU and the hardware dispatcher balances the load
//     (and staggers the waves' phases: a persistent grid runs them in lockstep and measured 14 % slower).
//   * only window rows/cols 5..60 are ever read by the learned patches, so only those 56x56 samples
//     are gathered (49 steps of an 8x8 lane tile instead of 64: one gather instruction touches a
//     ~11x11 pixel footprint of the L2-resident level, about a dozen cache lines).
//   * the window lives in LDS four times, shifted by 0..3 bytes, so that EVERY patch row (8 pixels
//     at an arbitrary byte offset) is a dword-aligned 8-byte LDS read (ds_read2_b32) from the copy
//     selected by (offset & 3).
//   * lane = triplet (8 rounds x 64 lanes): no cross-lane reduction.  With four pixels per dword,
//     S = sum(A*A) - sum(C*C) - 2*(sum(A*B) - sum(C*B)) is four v_dot4_u32_u8 per dword triple,
//     i.e. ONE VALU op per pixel-test.
//   * WHICH triplet a (round, lane) slot evaluates is chosen offline (tools/opt_latch_layout.py) so
//     that the 32 lanes of a half-wave hit distinct LDS banks as far as the learned pattern allows
//     (average conflict degree 3.4 -> 1.7); one ds_bpermute per output round puts the sign bits
//     back into descriptor order before the wave ballot that forms two output words.
//
// Bound: LDS reads (48 dwords per test) and the window gather; compulsory HBM is 20 B in + 64 B
// out per keypoint.
#include "clc_internal.h"
#include "clc_sincos.h"
#include "latch_layout.inc"
#include "latch_layout_swap.inc"

namespace clc {

static constexpr int kRow0 = LATCH_ROW0, kCol0 = LATCH_COL0;   // first window row/col kept in LDS
static constexpr int kStride = LATCH_STRIDE;                   // bytes per stored window row
static constexpr int kWinDwords = LATCH_NROWS * LATCH_STRIDE / 4;
static constexpr int kWaveLds = LATCH_WAVE_BYTES;
static_assert(kWaveLds <= 0x8000, "bit 15 of a slot record's first LDS address carries the role-exchange flag (make_slot_table)");
static constexpr int kCopyBase[4] = LATCH_COPY_BASES;
static constexpr int kTile0 = 5;                               // learned patches cover rows/cols 5..60
static constexpr int kTiles = 7;

struct PatchRow { uint8_t v[6]; };
static constexpr PatchRow k_pattern[512] = {
#include "latch_pattern.inc"
};
// (round, lane) -> learned triplet.  Round 4: the assignment searched WITH role exchange (tools/latch_anneal.c mode 0, swap 1) -- a slot may
// read its triplet's a and c patches the other way round, which gives the bank assignment more freedom: conflict degree sum 81 -> 77.
// Such a slot computes -S; its bit is taken as S' > 0.  -DCLC_CLATCH_NO_SWAP: round 3's assignment (A/B runs).
#ifdef CLC_CLATCH_NO_SWAP
static constexpr uint16_t k_slot_triplet[512] = LATCH_SLOT_TRIPLET;
#else
static constexpr uint16_t k_slot_triplet[512] = LATCH_SLOT_TRIPLET_EX;
#endif

// LDS byte address (inside the wave's region) of the dword-aligned start of the patch whose
// top-left window pixel is (row, col): copy k = (p & 3) holds win[i + k] at byte i.
static constexpr uint16_t patch_lds_addr(int row, int col)
{
    const int p = (row - kRow0) * kStride + (col - kCol0);
    return (uint16_t)(kCopyBase[p & 3] + (p & ~3));
}
struct SlotTable {
    // rec[round*64 + lane] = {a, b, c, src}: LDS addresses of the slot's patches a, b, c, and -- for the OUTPUT
    // bit n = round*64 + lane -- where that bit was computed: (source lane * 4) | (source round << 8).
    // 8 bytes per record: one global_load_dwordx2 per round and lane.
    uint16_t rec[512][4];
};
static constexpr SlotTable make_slot_table()
{
    SlotTable t{};
    for (int slot = 0; slot < 512; ++slot) {
        const int n = k_slot_triplet[slot] & 511;
        const bool sw = (k_slot_triplet[slot] >> 10) & 1;
        const uint16_t a = patch_lds_addr(k_pattern[n].v[0], k_pattern[n].v[1]), c = patch_lds_addr(k_pattern[n].v[4], k_pattern[n].v[5]);
        t.rec[slot][0] = (uint16_t)((sw ? c : a) | (sw ? 0x8000u : 0u));      // bit 15: roles exchanged (LDS addresses are < 2^15)
        t.rec[slot][1] = patch_lds_addr(k_pattern[n].v[2], k_pattern[n].v[3]);
        t.rec[slot][2] = sw ? a : c;
        t.rec[n][3] = (uint16_t)(((slot & 63) << 2) | ((slot >> 6) << 8));
    }
    return t;
}
__device__ __attribute__((aligned(16))) const SlotTable k_slots = make_slot_table();

typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 8 bytes at a dword-aligned LDS address (gfx950 executes a 4-byte-aligned ds_read_b64 at full rate; a
// byte-misaligned one is legal but was measured 4.6x slower end to end, hence the shifted copies)
__device__ __forceinline__ u32x2_a4 lds_read8(const uint8_t* p)
{
    return *reinterpret_cast<const u32x2_a4*>(p);
}

struct ClatchArgs {
    PyramidDesc pd;
    uint32_t slot_stride;                 // bytes between the pyramids of consecutive cameras (blockIdx.y = camera)
    const uint32_t* n_dev[kMaxBatch];     // after the GPU detector: a camera's keypoint count in device memory (nullable per camera)
    ClatchBatch cam;
};

__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot4(a, b, c, false);
}
__device__ __forceinline__ int clamp_i32(int v, int hi)   // min(max(v, 0), hi) in one v_med3_i32
{
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(hi));
    return r;
}

__global__ __launch_bounds__(64) void clatch_kernel(const ClatchArgs args, const uint8_t* __restrict__ arena_base)
{
    // One keypoint per wave, no loop (the grid is the keypoint count).  Round 3 (tools/archive/clatch_lab.hip, in-kernel stamps): a wave
    // used to spend 4.7 k of its 21.5 k cycles before its first gather -- it waited for its eight slot-table loads, THEN fetched
    // the keypoint with a vector load, THEN the level -- so: the table loads are issued first and consumed after the fill (they
    // are only needed by the tests), the keypoint comes through the scalar unit, and the copy phase is straight-line code with
    // immediate offsets.  Same bits; 147 -> 89 VGPRs; 69.2 -> 66.8 us per 2 x 10k launch on the same device.
    const uint32_t cam = blockIdx.y;
    const int n_arg = args.cam.n[cam];
    // keypoint count: a launch argument, or (after the GPU detector) read from device memory
    const uint32_t* __restrict__ n_dev = args.n_dev[cam];
    const int n = n_dev ? min((int)*n_dev, n_arg) : n_arg;
    const int kp = (int)blockIdx.x;
    if (kp >= n) return;
    __shared__ __attribute__((aligned(16))) uint8_t roi[kWaveLds];
    const uint32_t lane = threadIdx.x;

    // this lane's 8 slot records {a, b, c, src} (round j, lane): in flight during the whole fill
    uint2 rec[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rec[j] = *reinterpret_cast<const uint2*>(k_slots.rec[j * 64 + lane]);

    // the keypoint is wave-uniform: five dwords through the scalar unit {x, y, score | pad, angle, scale | pad}
    const uint32_t* __restrict__ kw = reinterpret_cast<const uint32_t*>(args.cam.kps[cam]) + (size_t)kp * 5u;
    const int px = (int)__builtin_amdgcn_readfirstlane(kw[0]);
    const int py = (int)__builtin_amdgcn_readfirstlane(kw[1]);
    const float angle = __uint_as_float(__builtin_amdgcn_readfirstlane(kw[3]));
    const int scale = (int)(__builtin_amdgcn_readfirstlane(kw[4]) & 0xFFu);
    static_assert(sizeof(clc_keypoint) == 20 && offsetof(clc_keypoint, angle) == 12 && offsetof(clc_keypoint, scale) == 16, "keypoint wire format");
    uint64_t* __restrict__ desc = args.cam.desc[cam];
    const uint8_t* __restrict__ arena = arena_base + (size_t)cam * args.slot_stride;
    const int lv = min(scale, args.pd.levels - 1);
    const LevelDesc L = args.pd.lv[lv];
    const uint8_t* __restrict__ img = arena + L.offset;
    // (sin, cos) in fp64 per wave costs a few % of this kernel in isolation; preparing it in another launch and loading it here
    // measured no better in round 2: the load sits at the head of the same dependency chain.
    float s, c;
    clc_sincosf(angle, &s, &c);
    const float fpx = (float)px, fpy = (float)py;
    const int wmax = (int)L.w - 1, hmax = (int)L.h - 1;
    const int dx = (int)(lane & 7u), dy = (int)(lane >> 3);

    // ---- window fill: tile (by, bx) covers rows 5+8*by.., cols 5+8*bx.. with an 8x8 lane tile
    // Round 5: the sample coordinates as PACKED fp32 pairs (v_pk_add_f32: two IEEE additions per instruction, the same roundings as two
    // v_add_f32 -- nothing is fused, the file is built with contraction off): {x, y} = ({fpx, fpy} + ({xo c, xo s} + {-(yo s), yo c})) + 0.5
    // is three instructions per sample instead of six.  a - b == a + (-b) in IEEE arithmetic, so the x coordinate's subtraction
    // (CLATCH.cu:166: xo*c - yo*s) keeps its bits.  The kernel issues vector instructions 77 % of the time (PMC, profiles/r05_clatch_notes.txt):
    // these 147 instructions per keypoint are a tenth of them.
    f32x2 XT[kTiles], YT[kTiles];
#pragma unroll
    for (int b = 0; b < kTiles; ++b) {
        const float xo = (float)(kTile0 + b * 8 + dx - 32);
        const float yo = (float)(kTile0 + b * 8 + dy - 32);
        XT[b] = f32x2{ xo * c, xo * s };
        YT[b] = f32x2{ -(yo * s), yo * c };
    }
    const f32x2 fp = { fpx, fpy }, half2 = { 0.5f, 0.5f };
    // (A clamp-free variant for keypoints >= 42 px inside the level was measured 17 % SLOWER in round 1 -- the second copy of the
    // unrolled fill costs more in instruction fetch than the two v_med3 save; round 3's stamps put the clamps at 0.5 k of the
    // fill's 7 k cycles.)
#pragma unroll
    for (int by = 0; by < kTiles; ++by) {
#pragma unroll
        for (int bx = 0; bx < kTiles; ++bx) {
            const f32x2 f = (fp + (XT[bx] + YT[by])) + half2;     // CLATCH.cu:166
            const float fx = f.x, fy = f.y;
            const int sx = clamp_i32((int)fx, wmax);
            const int sy = clamp_i32((int)fy, hmax);
            const uint32_t off = __umul24((uint32_t)sy, L.pitch) + (uint32_t)sx;
            roi[(kTile0 - kRow0 + by * 8 + dy) * kStride + (kTile0 - kCol0 + bx * 8 + dx)] = img[off];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- shifted copies 1..3: copy_k[i] = win[i + k]; all thirteen reads first, then the stores
    // (the 8-byte read of the window's last dword reaches 4 bytes past the window: those bytes lie in the gap in front of copy 1, are
    // never written, and only feed the three bytes past the end of each copy, which no patch row reaches -- a patch's last row ends
    // at window byte 4 * kWinDwords - 1 at the latest)
    static_assert(kCopyBase[1] >= 4 * kWinDwords + 8 && kCopyBase[2] >= kCopyBase[1] + 4 * kWinDwords + 8 &&
                  kCopyBase[3] >= kCopyBase[2] + 4 * kWinDwords + 8 && kWaveLds >= kCopyBase[3] + 4 * kWinDwords,
                  "the shifted copies must not overlap the bytes the copy phase reads past the window");
    {
        constexpr int kIters = (kWinDwords + 63) / 64;
        u32x2_a4 d[kIters];
#pragma unroll
        for (int k = 0; k < kIters; ++k) {
            const int i = (int)lane + 64 * k;
            if (k < kIters - 1 || i < kWinDwords) d[k] = *reinterpret_cast<const u32x2_a4*>(roi + 4 * i);
        }
#pragma unroll
        for (int k = 0; k < kIters; ++k) {
            const int i = (int)lane + 64 * k;
            if (k < kIters - 1 || i < kWinDwords) {
                *reinterpret_cast<uint32_t*>(roi + kCopyBase[1] + 4 * i) = __builtin_amdgcn_alignbyte(d[k].y, d[k].x, 1);
                *reinterpret_cast<uint32_t*>(roi + kCopyBase[2] + 4 * i) = __builtin_amdgcn_alignbyte(d[k].y, d[k].x, 2);
                *reinterpret_cast<uint32_t*>(roi + kCopyBase[3] + 4 * i) = __builtin_amdgcn_alignbyte(d[k].y, d[k].x, 3);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- 512 tests, 8 per lane
    uint32_t bits8 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t pa = rec[j].x & 0x7FFFu, pb = rec[j].x >> 16, pc = rec[j].y & 0xFFFFu;
        uint32_t aa = 0, cc = 0, ab = 0, cb = 0;
#pragma unroll
        for (int row = 0; row < 8; ++row) {
            const u32x2_a4 A = lds_read8(roi + pa + row * kStride);
            const u32x2_a4 B = lds_read8(roi + pb + row * kStride);
            const u32x2_a4 C = lds_read8(roi + pc + row * kStride);
            aa = udot4(A.x, A.x, aa); aa = udot4(A.y, A.y, aa);
            cc = udot4(C.x, C.x, cc); cc = udot4(C.y, C.y, cc);
            ab = udot4(A.x, B.x, ab); ab = udot4(A.y, B.y, ab);
            cb = udot4(C.x, B.x, cb); cb = udot4(C.y, B.y, cb);
        }
        const int32_t S = ((int32_t)aa - (int32_t)cc) - 2 * ((int32_t)ab - (int32_t)cb);
        const int32_t Se = (rec[j].x & 0x8000u) ? -S : S;                  // a slot that read a and c in exchanged roles holds -S
        bits8 |= (Se < 0 ? 1u : 0u) << j;
    }
    // ---- back to descriptor order: output round j, lane l <- bit of triplet 64*j + l
    // all eight gathers in flight at once; the sixteen dwords of the descriptor are then dealt to lanes 0..15 with v_writelane
    // (a ballot lives in a scalar register pair) and leave as ONE 64-byte store -- the select chain this replaces (a v_cmp and two
    // v_cndmask per word) and its pairwise waits were a third of this stage
    uint32_t got[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) got[j] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((rec[j].y >> 16) & 0xFFu), (int)bits8);
    uint32_t word = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint64_t bits = __ballot((got[j] >> (rec[j].y >> 24)) & 1u);
        // scalar data into lanes 2 j and 2 j + 1.  gfx940-family hazard: a VALU instruction that reads an SGPR the VALU has just
        // written (the v_cmp behind __ballot) needs two wait states; the compiler pads that for instructions it schedules, not
        // inside an asm statement (without the s_nop lane 2 j received the PREVIOUS word's bits)
        asm("s_nop 1\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
            : "+v"(word) : "s"((uint32_t)bits), "s"((uint32_t)(bits >> 32)), "n"(2 * j), "n"(2 * j + 1));
    }
    if (lane < 16u) reinterpret_cast<uint32_t*>(desc)[(size_t)kp * 16u + lane] = word;
}

static hipError_t launch_clatch_impl(const PyramidDesc& pd, const uint8_t* arena, size_t slot_stride, const ClatchBatch& batch,
                                     int n_img, const uint32_t* const* d_count, hipStream_t stream, Profiler* prof)
{
    if (n_img <= 0) return hipSuccess;
    if (n_img > kMaxBatch || slot_stride > 0xFFFFFFFFull) return hipErrorInvalidValue;
    ClatchArgs a;
    a.pd = pd;
    a.slot_stride = (uint32_t)slot_stride;
    int max_n = 0;
    for (int b = 0; b < kMaxBatch; ++b) {
        a.n_dev[b] = (d_count && b < n_img) ? d_count[b] : nullptr;
        a.cam.kps[b] = b < n_img ? batch.kps[b] : nullptr;
        a.cam.desc[b] = b < n_img ? batch.desc[b] : nullptr;
        a.cam.n[b] = b < n_img ? batch.n[b] : 0;
        if (a.cam.n[b] > max_n) max_n = a.cam.n[b];
    }
    if (max_n <= 0) return hipSuccess;
    const int blocks = max_n;                           // one wave per keypoint (gridDim.x reaches 2^31 - 1)
    prof_mark(prof, CLC_KERNEL_CLATCH, true, stream);
    hipLaunchKernelGGL(clatch_kernel, dim3(blocks, (uint32_t)n_img), dim3(64), 0, stream, a, arena);
    prof_mark(prof, CLC_KERNEL_CLATCH, false, stream);
    return hipGetLastError();
}

hipError_t launch_clatch_batch(const PyramidDesc& pd, const uint8_t* arena, size_t slot_stride, const ClatchBatch& batch,
                               int n_img, hipStream_t stream, Profiler* prof)
{
    return launch_clatch_impl(pd, arena, slot_stride, batch, n_img, nullptr, stream, prof);
}

hipError_t launch_clatch(const PyramidDesc& pd, const uint8_t* arena, const clc_keypoint* d_kps, int n,
                         uint64_t* d_desc, hipStream_t stream, Profiler* prof)
{
    ClatchBatch one{};
    one.kps[0] = d_kps; one.desc[0] = d_desc; one.n[0] = n;
    return launch_clatch_impl(pd, arena, 0, one, 1, nullptr, stream, prof);
}

hipError_t launch_clatch_counted(const PyramidDesc& pd, const uint8_t* arena, const clc_keypoint* d_kps,
                                 const uint32_t* d_count, int max_n, uint64_t* d_desc, hipStream_t stream, Profiler* prof)
{
    ClatchBatch one{};
    one.kps[0] = d_kps; one.desc[0] = d_desc; one.n[0] = max_n;
    return launch_clatch_impl(pd, arena, 0, one, 1, &d_count, stream, prof);
}

hipError_t launch_clatch_counted_batch(const PyramidDesc& pd, const uint8_t* arena, size_t slot_stride, const ClatchBatch& batch,
                                       const uint32_t* const* d_count, int n_img, hipStream_t stream, Profiler* prof)
{
    return launch_clatch_impl(pd, arena, slot_stride, batch, n_img, d_count, stream, prof);
}

} // namespace clc
