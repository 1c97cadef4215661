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
//   * one WAVE64 per keypoint, 4 independent waves per workgroup, persistent over keypoints --
//     no workgroup barrier anywhere.
//   * window fill: 64 steps of an 8x8 lane tile, so one gather instruction touches a ~11x11 pixel
//     footprint (about a dozen cache lines) of the L2-resident level instead of a 64-pixel line.
//   * the window lives in LDS four times, shifted by 0..3 bytes, so that EVERY patch row (8
//     pixels at an arbitrary byte offset) is a dword-aligned 8-byte LDS read from the copy
//     selected by (offset & 3).  The three extra copies cost 18 wave-iterations of
//     ds_read2 + 3 v_alignbyte + 3 ds_write per keypoint.
//   * lane = triplet (8 triplets per lane): no cross-lane reduction at all.  With four pixels per
//     dword, S = sum(A*A) - sum(C*C) - 2*(sum(A*B) - sum(C*B)) is four v_dot4_u32_u8 per dword
//     triple, i.e. ONE VALU op per pixel-test instead of ~6.  The 64 sign bits of one round are a
//     single wave ballot = two output words.
//
// Bound: LDS reads (48 dwords per test, random banks) and the window gather; compulsory HBM is
// 20 B in + 64 B out per keypoint.
#include "clc_internal.h"
#include "clc_sincos.h"

namespace clc {

static constexpr int kRoiStride = 72;                 // reference ROI row stride (CLATCH.cu:158)
static constexpr int kCopyBytes = 64 * kRoiStride + 32; // 4640: one shifted copy (+ slack for the +3 shift)
static constexpr int kCopies = 4;
static constexpr int kClatchWaves = 4;
static constexpr int kWaveLds = kCopies * kCopyBytes;   // 18560 B per wave, 74240 B per workgroup

struct PatchRow { uint8_t v[6]; };
static constexpr PatchRow k_pattern[512] = {
#include "latch_pattern.inc"
};

// LDS byte address (inside one wave's region) of the dword-aligned start of a patch whose top-left
// ROI byte offset is p = row*72 + col: copy (p & 3) holds roi[i + (p & 3)] at byte i.
struct PatchTable { uint16_t a[512][4]; };
static constexpr uint16_t patch_lds_addr(int row, int col)
{
    const int p = row * kRoiStride + col;
    return (uint16_t)((p & 3) * kCopyBytes + (p & ~3));
}
static constexpr PatchTable make_patch_table()
{
    PatchTable t{};
    for (int n = 0; n < 512; ++n) {
        t.a[n][0] = patch_lds_addr(k_pattern[n].v[0], k_pattern[n].v[1]);
        t.a[n][1] = patch_lds_addr(k_pattern[n].v[2], k_pattern[n].v[3]);
        t.a[n][2] = patch_lds_addr(k_pattern[n].v[4], k_pattern[n].v[5]);
        t.a[n][3] = 0;
    }
    return t;
}
__device__ const PatchTable k_patch_table = make_patch_table();

typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

struct ClatchArgs {
    PyramidDesc pd;
};

__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot4(a, b, c, false);
}

__global__ __launch_bounds__(64 * kClatchWaves) void clatch_kernel(const ClatchArgs args,
                                                                    const uint8_t* __restrict__ arena,
                                                                    const clc_keypoint* __restrict__ kps,
                                                                    const int n, uint64_t* __restrict__ desc)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kClatchWaves * kWaveLds];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t* const roi = lds + wave * kWaveLds;

    // this lane's 8 triplets: n = j*64 + lane
    uint32_t pa[8], pb[8], pc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint16_t* e = k_patch_table.a[j * 64 + lane];
        pa[j] = e[0]; pb[j] = e[1]; pc[j] = e[2];
    }
    const int dx = (int)(lane & 7u), dy = (int)(lane >> 3);

    for (int kp = (int)(blockIdx.x * kClatchWaves + wave); kp < n; kp += (int)(gridDim.x * kClatchWaves)) {
        const clc_keypoint pt = kps[kp];                      // wave-uniform
        const int lv = min((int)pt.scale, args.pd.levels - 1);
        const LevelDesc L = args.pd.lv[lv];
        const uint8_t* __restrict__ img = arena + L.offset;
        float s, c;
        clc_sincosf(pt.angle, &s, &c);
        const float fpx = (float)pt.x, fpy = (float)pt.y;
        const int wmax = (int)L.w - 1, hmax = (int)L.h - 1;

        // ---- window fill: step (by, bx) covers rows by*8.., cols bx*8.. with an 8x8 lane tile
        float xc[8], xs[8], ys[8], yc[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const float xo = (float)(b * 8 + dx - 32);
            const float yo = (float)(b * 8 + dy - 32);
            xc[b] = xo * c; xs[b] = xo * s;
            ys[b] = yo * s; yc[b] = yo * c;
        }
#pragma unroll
        for (int by = 0; by < 8; ++by) {
#pragma unroll
            for (int bx = 0; bx < 8; ++bx) {
                const float fx = (fpx + (xc[bx] - ys[by])) + 0.5f;   // CLATCH.cu:166
                const float fy = (fpy + (xs[bx] + yc[by])) + 0.5f;
                int sx = (int)fx, sy = (int)fy;
                sx = min(max(sx, 0), wmax);
                sy = min(max(sy, 0), hmax);
                const uint8_t v = img[(uint32_t)sy * L.pitch + (uint32_t)sx];
                roi[(by * 8 + dy) * kRoiStride + bx * 8 + dx] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- shifted copies 1..3: copy_s[i] = roi[i + s]
#pragma unroll 2
        for (int i = (int)lane; i < (64 * kRoiStride) / 4; i += 64) {
            const u32x2_a4 d = *reinterpret_cast<const u32x2_a4*>(roi + 4 * i);
            uint32_t* w1 = reinterpret_cast<uint32_t*>(roi + 1 * kCopyBytes + 4 * i);
            uint32_t* w2 = reinterpret_cast<uint32_t*>(roi + 2 * kCopyBytes + 4 * i);
            uint32_t* w3 = reinterpret_cast<uint32_t*>(roi + 3 * kCopyBytes + 4 * i);
            *w1 = __builtin_amdgcn_alignbyte(d.y, d.x, 1);
            *w2 = __builtin_amdgcn_alignbyte(d.y, d.x, 2);
            *w3 = __builtin_amdgcn_alignbyte(d.y, d.x, 3);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- 512 tests, 8 per lane; one ballot = 64 descriptor bits
        uint64_t mine = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t aa = 0, cc = 0, ab = 0, cb = 0;
#pragma unroll
            for (int row = 0; row < 8; ++row) {
                const u32x2_a4 A = *reinterpret_cast<const u32x2_a4*>(roi + pa[j] + row * kRoiStride);
                const u32x2_a4 B = *reinterpret_cast<const u32x2_a4*>(roi + pb[j] + row * kRoiStride);
                const u32x2_a4 C = *reinterpret_cast<const u32x2_a4*>(roi + pc[j] + row * kRoiStride);
                aa = udot4(A.x, A.x, aa); aa = udot4(A.y, A.y, aa);
                cc = udot4(C.x, C.x, cc); cc = udot4(C.y, C.y, cc);
                ab = udot4(A.x, B.x, ab); ab = udot4(A.y, B.y, ab);
                cb = udot4(C.x, B.x, cb); cb = udot4(C.y, B.y, cb);
            }
            const int32_t S = ((int32_t)aa - (int32_t)cc) - 2 * ((int32_t)ab - (int32_t)cb);
            const uint64_t bits = __ballot(S < 0);
            if (lane == (uint32_t)j) mine = bits;
        }
        if (lane < 8u) desc[(size_t)kp * 8u + lane] = mine;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

hipError_t launch_clatch(const PyramidDesc& pd, const uint8_t* arena, const clc_keypoint* d_kps, int n,
                         uint64_t* d_desc, hipStream_t stream, Profiler* prof)
{
    if (n <= 0) return hipSuccess;
    ClatchArgs a;
    a.pd = pd;
    int blocks = (n + kClatchWaves - 1) / kClatchWaves;
    if (blocks > 512) blocks = 512;   // 2 workgroups (74 KB LDS each) per CU x 256 CUs, persistent
    prof_mark(prof, CLC_KERNEL_CLATCH, true, stream);
    hipLaunchKernelGGL(clatch_kernel, dim3(blocks), dim3(64 * kClatchWaves), 0, stream, a, arena, d_kps, n, d_desc);
    prof_mark(prof, CLC_KERNEL_CLATCH, false, stream);
    return hipGetLastError();
}

} // namespace clc
