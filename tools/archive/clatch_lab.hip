// clatch_lab.hip -- round-3 laboratory for clatch_kernel: ablations (timing only, results wrong by construction) and
// bit-exact variants, one binary, every variant timed on the same keypoints (10 000 and 20 000, like bench.py's launch).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o tools/bin/clatch_lab tools/clatch_lab.hip
// The production kernel is included as is (variant "production"); the lab kernel below is a copy with switches.
#include "../coloc_amd/csrc/clatch.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
namespace clc { void prof_mark(Profiler*, int, bool, hipStream_t) {} }
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

namespace lab {
using namespace clc;

enum : unsigned {
    F_NOCONF     = 1u << 0,   // ablation: conflict-free patch addresses
    F_B64ALL     = 1u << 1,   // ablation: every patch row as one 8-byte-aligned ds_read_b64
    F_B64HALF    = 1u << 2,   // ablation: half of the (round, kind) pairs as aligned ds_read_b64
    F_NOCLAMP    = 1u << 3,   // ablation (interior keypoints only!): no v_med3 on the sample coordinates
    F_NOOUT      = 1u << 4,   // ablation: no un-permute (bits8 stored as they are)
    F_COPYUNROLL = 1u << 5,   // bit-exact: copy phase fully unrolled, reads first
    F_RPI        = 1u << 6,   // ablation unless proven: v_cvt_rpi_i32_f32(t) instead of (int)(t + 0.5f)
    F_NOSINCOS   = 1u << 7,   // ablation: s, c from cheap fp32 arithmetic
    F_NOFILL     = 1u << 8,   // ablation: no gather loads, coordinates only
    F_NOTEST     = 1u << 9,   // ablation: no tests
    F_STAMP      = 1u << 10,  // diagnostic: per-wave phase stamps (s_memrealtime / s_memtime) + HW_ID / XCC_ID into a side buffer
};

__device__ uint64_t* g_stamps;   // 16 uint64 per keypoint (F_STAMP builds only)
template <unsigned F>
__global__ __launch_bounds__(64) void clatch_lab_kernel(const ClatchArgs args, const uint8_t* __restrict__ arena_base)
{
    uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (F & F_STAMP) { st[0] = __builtin_amdgcn_s_memrealtime(); st[1] = __builtin_amdgcn_s_memtime(); }
    const uint32_t cam = blockIdx.y;
    const int n = args.cam.n[cam];
    if ((int)blockIdx.x >= n) return;
    const clc_keypoint* __restrict__ kps = args.cam.kps[cam];
    uint64_t* __restrict__ desc = args.cam.desc[cam];
    const uint8_t* __restrict__ arena = arena_base + (size_t)cam * args.slot_stride;
    __shared__ __attribute__((aligned(16))) uint8_t roi[kWaveLds];
    const uint32_t lane = threadIdx.x;

    uint32_t pa[8], pb[8], pc[8], src[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint2 e = *reinterpret_cast<const uint2*>(k_slots.rec[j * 64 + lane]);
        pa[j] = e.x & 0xFFFFu; pb[j] = e.x >> 16; pc[j] = e.y & 0xFFFFu;
        src[j] = e.y >> 16;
        if (F & F_NOCONF) {   // lane-linear, 8 bytes per lane: no two lanes of a half-wave on one bank
            pa[j] = lane * 8u + (uint32_t)j * 16u; pb[j] = 3192u + lane * 8u + (uint32_t)j * 16u; pc[j] = 6376u + lane * 8u + (uint32_t)j * 16u;
        }
        if (F & (F_B64ALL | F_B64HALF)) { pa[j] &= ~7u; pb[j] &= ~7u; pc[j] &= ~7u; }
    }
    const int dx = (int)(lane & 7u), dy = (int)(lane >> 3);

    for (int kp = (int)blockIdx.x; kp < n; kp += (int)gridDim.x) {
        const clc_keypoint pt = kps[kp];
        const int lv = min((int)pt.scale, args.pd.levels - 1);
        const LevelDesc L = args.pd.lv[lv];
        const uint8_t* __restrict__ img = arena + L.offset;
        float s, c;
        if (F & F_NOSINCOS) { s = pt.angle * 0.25f; c = 1.0f - s * s; }
        else clc_sincosf(pt.angle, &s, &c);
        const float fpx = (float)pt.x, fpy = (float)pt.y;
        const int wmax = (int)L.w - 1, hmax = (int)L.h - 1;

        float xc[kTiles], xs[kTiles], ys[kTiles], yc[kTiles];
#pragma unroll
        for (int b = 0; b < kTiles; ++b) {
            const float xo = (float)(kTile0 + b * 8 + dx - 32);
            const float yo = (float)(kTile0 + b * 8 + dy - 32);
            xc[b] = xo * c; xs[b] = xo * s;
            ys[b] = yo * s; yc[b] = yo * c;
        }
        uint32_t acc_nofill = 0;
        if (F & F_STAMP) { asm volatile("" :: "v"(xc[0]), "v"(ys[6])); st[2] = __builtin_amdgcn_s_memtime(); }
#pragma unroll
        for (int by = 0; by < kTiles; ++by) {
#pragma unroll
            for (int bx = 0; bx < kTiles; ++bx) {
                int sx, sy;
                if (F & F_RPI) {
                    const float tx = fpx + (xc[bx] - ys[by]);
                    const float ty = fpy + (xs[bx] + yc[by]);
                    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(sx) : "v"(tx));
                    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(sy) : "v"(ty));
                } else {
                    const float fx = (fpx + (xc[bx] - ys[by])) + 0.5f;
                    const float fy = (fpy + (xs[bx] + yc[by])) + 0.5f;
                    sx = (int)fx; sy = (int)fy;
                }
                if (!(F & F_NOCLAMP)) { sx = clamp_i32(sx, wmax); sy = clamp_i32(sy, hmax); }
                const uint32_t off = __umul24((uint32_t)sy, L.pitch) + (uint32_t)sx;
                uint8_t v;
                if (F & F_NOFILL) { acc_nofill ^= off; v = (uint8_t)off; }
                else v = img[off];
                roi[(kTile0 - kRow0 + by * 8 + dy) * kStride + (kTile0 - kCol0 + bx * 8 + dx)] = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (F & F_STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st[3] = __builtin_amdgcn_s_memtime(); }

        if (F & F_COPYUNROLL) {
            constexpr int kIters = (kWinDwords + 63) / 64;   // 13
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
        } else {
#pragma unroll 2
            for (int i = (int)lane; i < kWinDwords; i += 64) {
                const u32x2_a4 d = *reinterpret_cast<const u32x2_a4*>(roi + 4 * i);
                *reinterpret_cast<uint32_t*>(roi + kCopyBase[1] + 4 * i) = __builtin_amdgcn_alignbyte(d.y, d.x, 1);
                *reinterpret_cast<uint32_t*>(roi + kCopyBase[2] + 4 * i) = __builtin_amdgcn_alignbyte(d.y, d.x, 2);
                *reinterpret_cast<uint32_t*>(roi + kCopyBase[3] + 4 * i) = __builtin_amdgcn_alignbyte(d.y, d.x, 3);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        if (F & F_STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st[4] = __builtin_amdgcn_s_memtime(); }
        uint32_t bits8 = acc_nofill & 1u;
        if (!(F & F_NOTEST)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint32_t aa = 0, cc = 0, ab = 0, cb = 0;
#pragma unroll
                for (int row = 0; row < 8; ++row) {
                    u32x2_a4 A, B, C;
                    constexpr bool all64 = (F & F_B64ALL) != 0;
                    const bool a64 = all64 || ((F & F_B64HALF) && (j & 1));
                    const bool b64 = all64 || ((F & F_B64HALF) && (j & 2));
                    const bool c64 = all64 || ((F & F_B64HALF) && ((j >> 2) ^ (j & 1)));
                    if (a64) { const uint2 t = *reinterpret_cast<const uint2*>(roi + pa[j] + row * kStride); A.x = t.x; A.y = t.y; }
                    else A = lds_read8(roi + pa[j] + row * kStride);
                    if (b64) { const uint2 t = *reinterpret_cast<const uint2*>(roi + pb[j] + row * kStride); B.x = t.x; B.y = t.y; }
                    else B = lds_read8(roi + pb[j] + row * kStride);
                    if (c64) { const uint2 t = *reinterpret_cast<const uint2*>(roi + pc[j] + row * kStride); C.x = t.x; C.y = t.y; }
                    else C = lds_read8(roi + pc[j] + row * kStride);
                    aa = udot4(A.x, A.x, aa); aa = udot4(A.y, A.y, aa);
                    cc = udot4(C.x, C.x, cc); cc = udot4(C.y, C.y, cc);
                    ab = udot4(A.x, B.x, ab); ab = udot4(A.y, B.y, ab);
                    cb = udot4(C.x, B.x, cb); cb = udot4(C.y, B.y, cb);
                }
                const int32_t S = ((int32_t)aa - (int32_t)cc) - 2 * ((int32_t)ab - (int32_t)cb);
                bits8 |= (S < 0 ? 1u : 0u) << j;
            }
        }
        if (F & F_STAMP) { asm volatile("" :: "v"(bits8)); st[5] = __builtin_amdgcn_s_memtime(); }
        if (F & F_NOOUT) {
            // 8 bits per lane stored as they are (wrong order by construction): 16 lanes x 4 bytes
            uint32_t w = bits8;
            w |= (uint32_t)__builtin_amdgcn_mov_dpp((int)bits8, 0x101 /* row_shl:1 */, 0xF, 0xF, true) << 8;
            w |= (uint32_t)__builtin_amdgcn_mov_dpp((int)bits8, 0x102, 0xF, 0xF, true) << 16;
            w |= (uint32_t)__builtin_amdgcn_mov_dpp((int)bits8, 0x103, 0xF, 0xF, true) << 24;
            if ((lane & 3u) == 0) reinterpret_cast<uint32_t*>(desc)[(size_t)kp * 16u + (lane >> 2)] = w;
        } else {
            uint64_t mine = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src[j] & 0xFFu), (int)bits8);
                const uint64_t bits = __ballot((got >> (src[j] >> 8)) & 1u);
                if (lane == (uint32_t)j) mine = bits;
            }
            if (lane < 8u) desc[(size_t)kp * 8u + lane] = mine;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (F & F_STAMP) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            st[6] = __builtin_amdgcn_s_memtime(); st[7] = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) {
                uint64_t* o = g_stamps + 16u * ((size_t)cam * (size_t)n + kp);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = st[i];
                o[8] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11));     // HW_ID
                o[9] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));     // XCC_ID
            }
        }
    }
}


// ---- v3: one keypoint per wave, no loop; scalar keypoint load; slot-table loads issued first and consumed after the fill;
//      copies unrolled with immediate offsets.  Bit-exact.
enum : unsigned { V3_STAMP = 1u, V3_NOCLAMP = 2u, V3_NOGATHER = 4u, V3_NOTEST = 8u, V3_NOSINCOS = 16u, V3_NOCOPY = 32u, V3_NOOUT = 64u, V3_NOCOORD = 128u, V3_G1 = 256u, V3_G64 = 512u, V3_GDW = 1024u,
                  V3_B64 = 2048u /* timing only: every patch row read as an 8-byte-aligned ds_read_b64 (address & ~7) */,
                  V3_ADDTID = 4096u /* bit-exact: the shifted copies stored with ds_write_addtid_b32 */,
                  V3_PRIO = 8192u /* bit-exact: s_setprio 3 while the fill issues its gathers, 0 from the copies on */,
                  V3_PRIO_INV = 16384u /* the other way round: tests at priority 3 */,
                  V3_TILED = 32768u /* timing only: gather addresses as if the level were stored in 2 x 2-pixel dwords */ };
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <unsigned V>
__global__ __launch_bounds__(64) void clatch_v3_kernel(const ClatchArgs args, const uint8_t* __restrict__ arena_base)
{
    uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (V & V3_STAMP) { st[0] = __builtin_amdgcn_s_memrealtime(); st[1] = __builtin_amdgcn_s_memtime(); }
    const uint32_t cam = blockIdx.y;
    const int n = args.cam.n[cam];
    const int kp = (int)blockIdx.x;
    if (kp >= n) return;
    __shared__ __attribute__((aligned(16))) uint8_t roi[kWaveLds];
    const uint32_t lane = threadIdx.x;
    // the lane's 8 slot records: in flight during the whole fill
    uint2 rec[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rec[j] = *reinterpret_cast<const uint2*>(k_slots.rec[j * 64 + lane]);

    // keypoint through the scalar unit: {x, y, score|pad, angle} + {scale|pad}
    const uint32_t* __restrict__ kw = reinterpret_cast<const uint32_t*>(args.cam.kps[cam]) + (size_t)kp * 5u;
    const int px = (int)__builtin_amdgcn_readfirstlane(kw[0]);
    const int py = (int)__builtin_amdgcn_readfirstlane(kw[1]);
    const float angle = __uint_as_float(__builtin_amdgcn_readfirstlane(kw[3]));
    const int scale = (int)(__builtin_amdgcn_readfirstlane(kw[4]) & 0xFFu);
    uint64_t* __restrict__ desc = args.cam.desc[cam];
    const uint8_t* __restrict__ arena = arena_base + (size_t)cam * args.slot_stride;
    const int lv = min(scale, args.pd.levels - 1);
    const LevelDesc L = args.pd.lv[lv];
    const uint8_t* __restrict__ img = arena + L.offset;
    float s, c;
    if (V & V3_NOSINCOS) { s = angle * 0.25f; c = 1.0f - s * s; }
    else clc_sincosf(angle, &s, &c);
    const float fpx = (float)px, fpy = (float)py;
    const int wmax = (int)L.w - 1, hmax = (int)L.h - 1;
    const int dx = (int)(lane & 7u), dy = (int)(lane >> 3);

    if (V & V3_PRIO) __builtin_amdgcn_s_setprio(3);
    float xc[kTiles], xs[kTiles], ys[kTiles], yc[kTiles];
#pragma unroll
    for (int b = 0; b < kTiles; ++b) {
        const float xo = (float)(kTile0 + b * 8 + dx - 32);
        const float yo = (float)(kTile0 + b * 8 + dy - 32);
        xc[b] = xo * c; xs[b] = xo * s;
        ys[b] = yo * s; yc[b] = yo * c;
    }
    if (V & V3_STAMP) { asm volatile("" :: "v"(xc[0]), "v"(ys[6])); st[2] = __builtin_amdgcn_s_memtime(); }
#pragma unroll
    for (int by = 0; by < kTiles; ++by) {
#pragma unroll
        for (int bx = 0; bx < kTiles; ++bx) {
            const float fx = (fpx + (xc[bx] - ys[by])) + 0.5f;
            const float fy = (fpy + (xs[bx] + yc[by])) + 0.5f;
            int sx = (int)fx, sy = (int)fy;
            if (!(V & V3_NOCLAMP)) { sx = clamp_i32(sx, wmax); sy = clamp_i32(sy, hmax); }
            uint32_t off = (V & V3_NOCOORD) ? (uint32_t)(by * 8 + dy) * L.pitch + (uint32_t)(bx * 8 + dx) : __umul24((uint32_t)sy, L.pitch) + (uint32_t)sx;
            if (V & V3_TILED) off = (((uint32_t)sy >> 1) * (L.pitch >> 1) + ((uint32_t)sx >> 1)) * 4u + (((uint32_t)sy & 1u) * 2u + ((uint32_t)sx & 1u));
            if (V & V3_G1) off = (uint32_t)(by * 8) * L.pitch + (uint32_t)(bx * 64) + lane;              // one 64-byte line per instruction
            if (V & V3_G64) off = (uint32_t)(by * 8 + (int)lane) * L.pitch + (uint32_t)(bx * 8);          // 64 lines per instruction
            if (V & V3_GDW) off = (uint32_t)(by * 8 + dy) * L.pitch + (uint32_t)(bx * 8 + dx) * 4u;       // 8 lines, lanes 4 bytes apart
            roi[(kTile0 - kRow0 + by * 8 + dy) * kStride + (kTile0 - kCol0 + bx * 8 + dx)] = (V & V3_NOGATHER) ? (uint8_t)off : img[off];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (V & V3_STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st[3] = __builtin_amdgcn_s_memtime(); }
    if (V & V3_PRIO) __builtin_amdgcn_s_setprio(0);
    if (V & V3_PRIO_INV) __builtin_amdgcn_s_setprio(3);
    if (!(V & V3_NOCOPY)) {
        constexpr int kIters = (kWinDwords + 63) / 64;   // 13
        u32x2_a4 d[kIters];
#pragma unroll
        for (int k = 0; k < kIters; ++k) {
            const int i = (int)lane + 64 * k;
            if (k < kIters - 1 || i < kWinDwords) d[k] = *reinterpret_cast<const u32x2_a4*>(roi + 4 * i);
        }
        if (V & V3_ADDTID) {
            // address = M0[15:0] + offset + 4 * lane: exactly the copies' pattern, and no address register travels to the LDS
            asm volatile("s_mov_b32 m0, %0" :: "s"((uint32_t)(uintptr_t)roi) : "memory");
        }
#pragma unroll
        for (int k = 0; k < kIters; ++k) {
            const int i = (int)lane + 64 * k;
            if (k < kIters - 1 || i < kWinDwords) {
                if (V & V3_ADDTID) {
#pragma unroll
                    for (int sh = 1; sh < 4; ++sh)
                        asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(__builtin_amdgcn_alignbyte(d[k].y, d[k].x, sh)), "n"(kCopyBase[sh] + 256 * k) : "memory");
                } else {
                *reinterpret_cast<uint32_t*>(roi + kCopyBase[1] + 4 * i) = __builtin_amdgcn_alignbyte(d[k].y, d[k].x, 1);
                *reinterpret_cast<uint32_t*>(roi + kCopyBase[2] + 4 * i) = __builtin_amdgcn_alignbyte(d[k].y, d[k].x, 2);
                *reinterpret_cast<uint32_t*>(roi + kCopyBase[3] + 4 * i) = __builtin_amdgcn_alignbyte(d[k].y, d[k].x, 3);
                }
            }
        }
        if (V & V3_ADDTID) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 m0, -1" ::: "memory");
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (V & V3_STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st[4] = __builtin_amdgcn_s_memtime(); }

    uint32_t bits8 = 0;
    if (V & V3_NOTEST) bits8 = *reinterpret_cast<const uint32_t*>(roi + 4 * lane) & 0xFFu;
    else
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t pa = rec[j].x & 0xFFFFu, pb = rec[j].x >> 16, pc = rec[j].y & 0xFFFFu;
        if (V & V3_B64) { pa &= ~7u; pb &= ~7u; pc &= ~7u; }
        uint32_t aa = 0, cc = 0, ab = 0, cb = 0;
#pragma unroll
        for (int row = 0; row < 8; ++row) {
            u32x2_a4 A, B, C;
            if (V & V3_B64) {
                const uint2 a8 = *reinterpret_cast<const uint2*>(roi + pa + row * kStride), b8 = *reinterpret_cast<const uint2*>(roi + pb + row * kStride),
                            c8 = *reinterpret_cast<const uint2*>(roi + pc + row * kStride);
                A.x = a8.x; A.y = a8.y; B.x = b8.x; B.y = b8.y; C.x = c8.x; C.y = c8.y;
            } else {
                A = lds_read8(roi + pa + row * kStride);
                B = lds_read8(roi + pb + row * kStride);
                C = lds_read8(roi + pc + row * kStride);
            }
            aa = udot4(A.x, A.x, aa); aa = udot4(A.y, A.y, aa);
            cc = udot4(C.x, C.x, cc); cc = udot4(C.y, C.y, cc);
            ab = udot4(A.x, B.x, ab); ab = udot4(A.y, B.y, ab);
            cb = udot4(C.x, B.x, cb); cb = udot4(C.y, B.y, cb);
        }
        const int32_t S = ((int32_t)aa - (int32_t)cc) - 2 * ((int32_t)ab - (int32_t)cb);
        bits8 |= (S < 0 ? 1u : 0u) << j;
    }
    if (V & V3_STAMP) { asm volatile("" :: "v"(bits8)); st[5] = __builtin_amdgcn_s_memtime(); }
    uint64_t mine = 0;
    if (V & V3_NOOUT) mine = bits8 ^ (rec[0].y >> 16);
    else
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t src = rec[j].y >> 16;
        const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src & 0xFFu), (int)bits8);
        const uint64_t bits = __ballot((got >> (src >> 8)) & 1u);
        if (lane == (uint32_t)j) mine = bits;
    }
    if (lane < 8u) desc[(size_t)kp * 8u + lane] = mine;
    if (V & V3_STAMP) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        st[6] = __builtin_amdgcn_s_memtime(); st[7] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            uint64_t* o = g_stamps + 16u * ((size_t)cam * (size_t)n + kp);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = st[i];
            o[8] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11));
            o[9] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));
        }
    }
}



// clc_sincosf with its constants read from a table (same operations in the same order, same constants): inside a loop the
// compiler otherwise hoists the twenty-odd v_mov pairs that materialise the fp64 literals and keeps them live for the whole kernel
__constant__ double k_sincos_tab[16] = {
    6.36619772367581382433e-01, 1.57079632673412561417e+00, 6.07710050650619224932e-11, 0.0,
    -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04, 2.75573137070700676789e-06,
    -2.50507602534068634195e-08, 1.58969099521155010221e-10,
    4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05, -2.75573143513906633035e-07,
    2.08757232129817482790e-09, -1.13596475577881948265e-11 };
__device__ __forceinline__ void sincos_tab(const float angle, const double* __restrict__ kt, float* s_out, float* c_out)
{
    const double x = (double)angle;
    const double kd0 = x * kt[0];
    const double kd1 = kd0 + (kd0 < 0.0 ? -0.5 : 0.5);
    const double kd2 = kd1 > 2.0e9 ? 2.0e9 : (kd1 < -2.0e9 ? -2.0e9 : kd1);
    const int k = (int)kd2;
    const double kd = (double)k;
    const double r = (x - kd * kt[1]) - kd * kt[2];
    const double z = r * r;
    double ps = kt[9];
    ps = __builtin_fma(ps, z, kt[8]);
    ps = __builtin_fma(ps, z, kt[7]);
    ps = __builtin_fma(ps, z, kt[6]);
    ps = __builtin_fma(ps, z, kt[5]);
    ps = __builtin_fma(ps, z, kt[4]);
    const double sr = __builtin_fma(r * z, ps, r);
    double pc = kt[15];
    pc = __builtin_fma(pc, z, kt[14]);
    pc = __builtin_fma(pc, z, kt[13]);
    pc = __builtin_fma(pc, z, kt[12]);
    pc = __builtin_fma(pc, z, kt[11]);
    pc = __builtin_fma(pc, z, kt[10]);
    const double cr = __builtin_fma(z * z, pc, 1.0 - 0.5 * z);
    double sv, cv;
    switch (k & 3) {
        case 0: sv = sr; cv = cr; break;
        case 1: sv = cr; cv = -sr; break;
        case 2: sv = -sr; cv = -cr; break;
        default: sv = -cr; cv = sr; break;
    }
    *s_out = (float)sv;
    *c_out = (float)cv;
}

// ---- pool: ONE persistent 16-wave workgroup per CU; the twelve 12.7 KB window regions of the CU are a pool: a wave holds one only from
//      the fill to the end of the tests (75 % of a keypoint's time), does its keypoint load / sincos / output without one, and takes the
//      next keypoint of the workgroup's range from an LDS counter.  Bit-exact.
enum : unsigned { P_STAMP = 1u };
static constexpr int kPoolSlots = 12;
static constexpr int kPoolWaves = 16;
template <unsigned V>
__global__ __launch_bounds__(64 * kPoolWaves) void clatch_pool_kernel(const ClatchArgs args, const uint8_t* __restrict__ arena_base)
{
    __shared__ __attribute__((aligned(16))) uint8_t pool[kPoolSlots * kWaveLds];
    __shared__ uint32_t s_free, s_next;
    const uint32_t lane = threadIdx.x & 63u;
    // flattened (camera, keypoint) items of this workgroup: an equal contiguous share
    uint32_t total = 0;
#pragma unroll
    for (int b = 0; b < kMaxBatch; ++b) total += (uint32_t)args.cam.n[b];
    const uint32_t begin = (uint32_t)(((uint64_t)total * blockIdx.x) / gridDim.x);
    const uint32_t end = (uint32_t)(((uint64_t)total * (blockIdx.x + 1u)) / gridDim.x);
    if (threadIdx.x == 0) { s_free = (1u << kPoolSlots) - 1u; s_next = begin; }
    __syncthreads();
    uint2 rec[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rec[j] = *reinterpret_cast<const uint2*>(k_slots.rec[j * 64 + lane]);
    const uint32_t lane_fixed = lane;

    for (;;) {
        // everything derived from the lane number is recomputed per keypoint (a few dozen cheap instructions): hoisted out of the
        // loop it would pin ~40 registers for the whole kernel (and spilled at the 128 the 16-wave workgroup allows)
        uint32_t lane = lane_fixed;
        asm volatile("" : "+v"(lane));
        const int dx = (int)(lane & 7u), dy = (int)(lane >> 3);
        uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (V & P_STAMP) { st[0] = __builtin_amdgcn_s_memrealtime(); st[1] = __builtin_amdgcn_s_memtime(); }
        uint32_t item = 0;
        if (lane_fixed == 0) item = __hip_atomic_fetch_add(&s_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= end) break;
        uint32_t cam = 0, kp = item;
#pragma unroll
        for (int b = 0; b < kMaxBatch - 1; ++b) if (cam == (uint32_t)b && kp >= (uint32_t)args.cam.n[b]) { kp -= (uint32_t)args.cam.n[b]; cam = b + 1; }
        const uint32_t* __restrict__ kw = reinterpret_cast<const uint32_t*>(args.cam.kps[cam]) + (size_t)kp * 5u;
        const int px = (int)__builtin_amdgcn_readfirstlane(kw[0]);
        const int py = (int)__builtin_amdgcn_readfirstlane(kw[1]);
        const float angle = __uint_as_float(__builtin_amdgcn_readfirstlane(kw[3]));
        const int scale = (int)(__builtin_amdgcn_readfirstlane(kw[4]) & 0xFFu);
        uint64_t* __restrict__ desc = args.cam.desc[cam];
        const uint8_t* __restrict__ arena = arena_base + (size_t)cam * args.slot_stride;
        const int lv = min(scale, args.pd.levels - 1);
        const LevelDesc L = args.pd.lv[lv];
        const uint8_t* __restrict__ img = arena + L.offset;
        float s, c;
        {
            const double* kt = k_sincos_tab;
            asm volatile("" : "+s"(kt));          // per keypoint: two scalar loads instead of 40 registers for the whole loop
            sincos_tab(angle, kt, &s, &c);
        }
        const float fpx = (float)px, fpy = (float)py;
        const int wmax = (int)L.w - 1, hmax = (int)L.h - 1;
        float xc[kTiles], xs[kTiles], ys[kTiles], yc[kTiles];
#pragma unroll
        for (int b = 0; b < kTiles; ++b) {
            const float xo = (float)(kTile0 + b * 8 + dx - 32);
            const float yo = (float)(kTile0 + b * 8 + dy - 32);
            xc[b] = xo * c; xs[b] = xo * s;
            ys[b] = yo * s; yc[b] = yo * c;
        }
        // ---- take a window region
        uint32_t slot = 0;
        if (lane == 0) {
            for (;;) {
                const uint32_t m = __hip_atomic_load(&s_free, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (m) {
                    const uint32_t bit = m & (0u - m);
                    const uint32_t old = __hip_atomic_fetch_and(&s_free, ~bit, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (old & bit) { slot = (uint32_t)__builtin_ctz(bit); break; }
                } else __builtin_amdgcn_s_sleep(4);
            }
        }
        slot = __builtin_amdgcn_readfirstlane(slot);
        uint8_t* roi = pool + slot * (uint32_t)kWaveLds;
        if (V & P_STAMP) { asm volatile("" :: "v"(xc[0]), "v"(ys[6])); st[2] = __builtin_amdgcn_s_memtime(); }
#pragma unroll
        for (int by = 0; by < kTiles; ++by) {
#pragma unroll
            for (int bx = 0; bx < kTiles; ++bx) {
                const float fx = (fpx + (xc[bx] - ys[by])) + 0.5f;
                const float fy = (fpy + (xs[bx] + yc[by])) + 0.5f;
                const int sx = clamp_i32((int)fx, wmax), sy = clamp_i32((int)fy, hmax);
                const uint32_t off = __umul24((uint32_t)sy, L.pitch) + (uint32_t)sx;
                roi[(kTile0 - kRow0 + by * 8 + dy) * kStride + (kTile0 - kCol0 + bx * 8 + dx)] = img[off];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (V & P_STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st[3] = __builtin_amdgcn_s_memtime(); }
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
        if (V & P_STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st[4] = __builtin_amdgcn_s_memtime(); }
        uint32_t bits8 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (the unpacked addresses must not be hoisted out of the keypoint loop: 24 registers for the whole loop)
            uint32_t rx = rec[j].x, ry = rec[j].y;
            asm volatile("" : "+v"(rx), "+v"(ry));
            const uint32_t pa = rx & 0xFFFFu, pb = rx >> 16, pc = ry & 0xFFFFu;
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
            bits8 |= (S < 0 ? 1u : 0u) << j;
            // one round's 24 reads in flight, not all eight rounds' (register budget 128): pin the round's arithmetic above this
            // point and the next round's reads below it
            asm volatile("" : "+v"(bits8) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // every read of the region has returned (its data was consumed above): hand it back
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_or(&s_free, 1u << slot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (V & P_STAMP) { asm volatile("" :: "v"(bits8)); st[5] = __builtin_amdgcn_s_memtime(); }
        uint64_t mine = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t ry = rec[j].y;
            asm volatile("" : "+v"(ry));
            const uint32_t src = ry >> 16;
            const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src & 0xFFu), (int)bits8);
            const uint64_t bits = __ballot((got >> (src >> 8)) & 1u);
            if (lane == (uint32_t)j) mine = bits;
        }
        if (lane < 8u) desc[(size_t)kp * 8u + lane] = mine;
        if (V & P_STAMP) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            st[6] = __builtin_amdgcn_s_memtime(); st[7] = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) {
                uint64_t* o = g_stamps + 16u * (size_t)item;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = st[i];
                o[8] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11));
                o[9] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));
            }
        }
    }
}

// exhaustive semantics check of v_cvt_rpi_i32_f32 against (int)(x + 0.5f) for every fp32 with |x| < 2^24
__global__ void rpi_check_kernel(unsigned long long* out)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t bad = 0, bad_pos = 0; uint32_t first = 0xFFFFFFFFu;
    for (uint64_t b = gid; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)b;
        const float x = __uint_as_float(bits);
        if (!(fabsf(x) < 16777216.0f)) continue;
        int r; asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
        const float u = x + 0.5f;
        const int w = (int)u;
        // what the kernel needs: equality after clamping to [0, hi] for any hi >= 0, i.e. equality of max(., 0)
        if (max(r, 0) != max(w, 0)) { ++bad; if (x >= 0.f) ++bad_pos; if (bits < first) first = bits; }
    }
    atomicAdd(&out[0], bad); atomicAdd(&out[1], bad_pos);
    atomicMin(reinterpret_cast<unsigned int*>(&out[2]), first);
}

static const char* g_filter = nullptr; static int g_only_n = 0; static int g_reps = 100;
static bool selected(const char* name, int n)
{
    if (g_only_n && n != g_only_n) return false;
    if (!g_filter) return true;
    std::string f(g_filter); size_t pos = 0;
    while (pos <= f.size()) { size_t e = f.find(',', pos); if (e == std::string::npos) e = f.size();
        if (e > pos && strstr(name, f.substr(pos, e - pos).c_str())) return true; pos = e + 1; }
    return false;
}
template <unsigned F>
static float time_variant(const char* name, const PyramidDesc& pd, const uint8_t* darena, const clc_keypoint* dk, int n, uint64_t* dd, bool production,
                          unsigned dyn_lds = 0, int grid = 0, int v3 = -1)
{
    if (!selected(name, n)) return 0.f;
    ClatchArgs a; a.pd = pd; a.slot_stride = 0; for (int b = 0; b < kMaxBatch; ++b) a.n_dev[b] = nullptr;
    for (int b = 0; b < kMaxBatch; ++b) { a.cam.kps[b] = nullptr; a.cam.desc[b] = nullptr; a.cam.n[b] = 0; }
    a.cam.kps[0] = dk; a.cam.desc[0] = dd; a.cam.n[0] = n;
    auto launch = [&]() {
        const int g = grid ? grid : n;
        if (v3 == 10) hipLaunchKernelGGL(clatch_pool_kernel<0>, dim3(grid ? grid : 256), dim3(64 * kPoolWaves), 0, 0, a, darena);
        else if (v3 >= 100) {
            switch (v3 - 100) {
#define V3CASE(X) case X: hipLaunchKernelGGL(clatch_v3_kernel<X>, dim3(g), dim3(64), dyn_lds, 0, a, darena); break;
                V3CASE(4) V3CASE(8) V3CASE(16) V3CASE(32) V3CASE(64) V3CASE(128) V3CASE(132) V3CASE(12) V3CASE(252) V3CASE(6) V3CASE(140) V3CASE(36) V3CASE(72) V3CASE(384) V3CASE(640) V3CASE(1152) V3CASE(392) V3CASE(1160) V3CASE(136) V3CASE(648) V3CASE(2048) V3CASE(2052) V3CASE(4096) V3CASE(6144) V3CASE(6148) V3CASE(8192) V3CASE(16384) V3CASE(32768) V3CASE(32776)
                default: printf("no such v3 variant\n"); exit(1);
            }
        }
        else if (v3 == 0) hipLaunchKernelGGL(clatch_v3_kernel<0>, dim3(g), dim3(64), dyn_lds, 0, a, darena);
        else if (v3 == 2) hipLaunchKernelGGL(clatch_v3_kernel<V3_NOCLAMP>, dim3(g), dim3(64), dyn_lds, 0, a, darena);
        else if (production) hipLaunchKernelGGL(clatch_kernel, dim3(g), dim3(64), dyn_lds, 0, a, darena);
        else hipLaunchKernelGGL(clatch_lab_kernel<F>, dim3(g), dim3(64), dyn_lds, 0, a, darena);
    };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipDeviceSynchronize());
    // sustained: 100 launches back to back, three times; the median of the three per-launch averages
    std::vector<float> ts;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < g_reps; ++i) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / (float)g_reps);
    }
    std::sort(ts.begin(), ts.end());
    std::vector<float> te;
    for (int i = 0; i < 21; ++i) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); te.push_back(ms); }
    std::sort(te.begin(), te.end());
    std::vector<uint64_t> hd((size_t)n * 8); CHECK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
    uint64_t chk = 0; for (auto v : hd) chk = chk * 1315423911ull + v;
    printf("  %-38s n=%5d sustained %7.2f us/launch | single %7.2f us (min %7.2f)  checksum %016llx\n", name, n, ts[1] * 1e3, te[10] * 1e3, te[0] * 1e3, (unsigned long long)chk);
    fflush(stdout);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ts[1] * 1e3f;
}

static void stamp_report(const PyramidDesc& pd, const uint8_t* darena, const clc_keypoint* dk, int n, uint64_t* dd, int which = 0)
{
    if (!selected(which == 10 ? "stamps pool" : which ? "stamps v3" : "stamps", n)) return;
    uint64_t* dst; CHECK(hipMalloc((void**)&dst, (size_t)n * 16 * 8)); CHECK(hipMemset(dst, 0, (size_t)n * 16 * 8));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst)));
    ClatchArgs a; a.pd = pd; a.slot_stride = 0; for (int b = 0; b < kMaxBatch; ++b) a.n_dev[b] = nullptr;
    for (int b = 0; b < kMaxBatch; ++b) { a.cam.kps[b] = nullptr; a.cam.desc[b] = nullptr; a.cam.n[b] = 0; }
    a.cam.kps[0] = dk; a.cam.desc[0] = dd; a.cam.n[0] = n;
    for (int i = 0; i < 5; ++i) {
        if (which == 10) hipLaunchKernelGGL(clatch_pool_kernel<P_STAMP>, dim3(256), dim3(64 * kPoolWaves), 0, 0, a, darena);
        else if (which == 1) hipLaunchKernelGGL(clatch_v3_kernel<V3_STAMP>, dim3(n), dim3(64), 0, 0, a, darena);
        else if (which == 2) hipLaunchKernelGGL(clatch_v3_kernel<V3_STAMP | V3_NOCLAMP>, dim3(n), dim3(64), 0, 0, a, darena);
        else hipLaunchKernelGGL(clatch_lab_kernel<F_STAMP>, dim3(n), dim3(64), 0, 0, a, darena);
    }
    CHECK(hipDeviceSynchronize());
    std::vector<uint64_t> h((size_t)n * 16); CHECK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipFree(dst));
    // phases (shader cycles)
    const char* names[5] = { "start -> sincos + products done", "fill (49 gathers landed, written)", "shifted copies", "512 tests", "un-permute, store" };
    double ph[5] = {0, 0, 0, 0, 0}, life = 0, life_rt = 0; uint64_t t0 = ~0ull, t1 = 0;
    for (int k = 0; k < n; ++k) { const uint64_t* o = &h[(size_t)k * 16];
        for (int i = 0; i < 5; ++i) ph[i] += (double)(o[i + 2] - o[i + 1]);
        life += (double)(o[6] - o[1]); life_rt += (double)(o[7] - o[0]); t0 = std::min(t0, o[0]); t1 = std::max(t1, o[7]); }
    printf("  stamped build %d, n=%d: kernel span %.1f us (first wave start -> last wave end, 100 MHz clock)\n", which, n, (t1 - t0) * 0.01);
    for (int i = 0; i < 5; ++i) printf("    %-36s %8.0f cycles  %4.1f %%\n", names[i], ph[i] / n, 100.0 * ph[i] / life);
    printf("    whole wave %8.0f cycles = %.2f us (realtime) -> clock %.2f GHz; sum of wave lives / span = %.2f waves resident per CU on average\n",
           life / n, life_rt / n * 0.01, life / life_rt * 0.1, life_rt / (double)(t1 - t0) / 256.0);
    // per-CU timeline: key = xcc | se | sh(?) | cu from HW_ID (cu_id[11:8], sh_id[12], se_id[15:13]) and XCC_ID
    struct Ev { uint64_t t; int d; };
    std::vector<std::vector<Ev>> cu(8 * 256);
    for (int k = 0; k < n; ++k) { const uint64_t* o = &h[(size_t)k * 16];
        const uint32_t hw = (uint32_t)o[8], xcc = (uint32_t)o[9] & 15u;
        const uint32_t key = (xcc << 8) | ((hw >> 8) & 0xFFu);
        cu[key].push_back({ o[0], +1 }); cu[key].push_back({ o[7], -1 }); }
    int ncu = 0; double gap_sum = 0; long gap_n = 0; std::vector<double> occ_hist(16, 0.0); double busy_span = 0;
    for (auto& v : cu) { if (v.empty()) continue; ++ncu;
        std::sort(v.begin(), v.end(), [](const Ev& x, const Ev& y) { return x.t < y.t || (x.t == y.t && x.d < y.d); });
        int c = 0; uint64_t last = v[0].t; uint64_t last_end = 0;
        for (auto& e : v) { occ_hist[std::min(c, 15)] += (double)(e.t - last); last = e.t;
            if (e.d < 0) last_end = e.t; else if (last_end && c >= 10) { gap_sum += (double)(e.t - last_end); ++gap_n; }
            c += e.d; }
        busy_span += (double)(v.back().t - v[0].t); }
    printf("    %d distinct CU keys; time share by waves resident on the CU:", ncu);
    double tot = 0; for (double x : occ_hist) tot += x;
    for (int i = 0; i < 16; ++i) if (occ_hist[i] > 0.002 * tot) printf("  %d:%.0f%%", i, 100.0 * occ_hist[i] / tot);
    printf("\n    mean gap between a wave ending and the next one starting on its CU (when >= 10 resident): %.2f us over %ld refills\n",
           gap_n ? gap_sum / gap_n * 0.01 : 0.0, gap_n);
}

} // namespace lab

int main(int argc, char** argv)
{
    using namespace clc; using namespace lab;
    if (argc > 1 && strlen(argv[1])) g_filter = argv[1];
    if (argc > 2) g_only_n = atoi(argv[2]);
    if (argc > 3) g_reps = atoi(argv[3]);
    const uint32_t W = 640, H = 480; const int NMAX = 20000;
    PyramidDesc pd{}; pd.levels = 8; float f = 1.f; uint32_t off = 0;
    for (int i = 0; i < 8; ++i) {
        if (i) f *= 1.2f;
        pd.lv[i].w = i ? (uint32_t)((float)W / f + 0.5f) : W; pd.lv[i].h = i ? (uint32_t)((float)H / f + 0.5f) : H;
        pd.lv[i].pitch = (pd.lv[i].w + 63) / 64 * 64; pd.lv[i].offset = off; off += (pd.lv[i].pitch * pd.lv[i].h + 255) / 256 * 256; pd.f[i] = f;
    }
    std::vector<uint8_t> harena(off + 256);
    uint64_t s = 88172645463325252ull; auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (auto& v : harena) v = (uint8_t)rnd();
    uint8_t* darena; clc_keypoint* dk; uint64_t* dd;
    CHECK(hipMalloc((void**)&darena, harena.size())); CHECK(hipMalloc((void**)&dk, NMAX * sizeof(clc_keypoint))); CHECK(hipMalloc((void**)&dd, (size_t)NMAX * 64));
    CHECK(hipMemcpy(darena, harena.data(), harena.size(), hipMemcpyHostToDevice));

    // --- v_cvt_rpi semantics
    if (!g_filter) {
        unsigned long long* dout; CHECK(hipMalloc((void**)&dout, 24));
        unsigned long long init[3] = { 0, 0, 0xFFFFFFFFull };
        CHECK(hipMemcpy(dout, init, 24, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(rpi_check_kernel, dim3(4096), dim3(256), 0, 0, dout);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[3]; CHECK(hipMemcpy(h, dout, 24, hipMemcpyDeviceToHost));
        printf("v_cvt_rpi_i32_f32 vs (int)(x+0.5f), clamped below at 0, all |x| < 2^24: %llu mismatches (%llu with x >= 0), first bits %08llx\n",
               h[0], h[1], h[2] & 0xFFFFFFFFull);
    }

    for (int interior = 0; interior < 2; ++interior) {
        std::vector<clc_keypoint> hk(NMAX);
        // area-weighted levels like tests/synth.random_keypoints: level l with probability ~ w*h
        double area[8], tot = 0; for (int i = 0; i < 8; ++i) { area[i] = (double)pd.lv[i].w * pd.lv[i].h; tot += area[i]; }
        for (auto& k : hk) {
            double u = (rnd() % 1000000) / 1e6 * tot; int l = 0; while (l < 7 && u >= area[l]) { u -= area[l]; ++l; }
            const int m = interior ? 45 : 3;
            k.scale = l; k.x = m + rnd() % (pd.lv[l].w - 2 * m); k.y = m + rnd() % (pd.lv[l].h - 2 * m);
            k.angle = ((int)(rnd() % 62832) - 31416) * 1e-4f; k.score = 0;
        }
        CHECK(hipMemcpy(dk, hk.data(), NMAX * sizeof(clc_keypoint), hipMemcpyHostToDevice));
        printf("== keypoints: %s, levels area-weighted\n", interior ? "INTERIOR (>= 45 px inside)" : "anywhere >= 3 px inside");
        for (int n : { 10000, 20000 }) {
            time_variant<0>("production", pd, darena, dk, n, dd, true);
            if (!interior) stamp_report(pd, darena, dk, n, dd);
            time_variant<0>("pool", pd, darena, dk, n, dd, false, 0, 0, 10);
            stamp_report(pd, darena, dk, n, dd, 10);
            time_variant<0>("v3", pd, darena, dk, n, dd, false, 0, 0, 0);
            if (!interior) {
                const struct { const char* nm; int v; } abl[] = {
                    { "v3abl: no gather loads", 4 }, { "v3abl: no tests", 8 }, { "v3abl: no sincos", 16 }, { "v3abl: no copies", 32 }, { "v3abl: no un-permute", 64 },
                    { "v3abl: no coordinates (gathers stay)", 128 }, { "v3abl: no coordinates, no gathers", 132 }, { "v3abl: no gathers, no tests", 12 },
                    { "v3abl: no copies, no tests", 36 + 4 }, { "v3abl: no tests, no un-permute", 72 }, { "v3abl: everything off", 252 },
                    { "v3abl: gathers within 1 line", 384 }, { "v3abl: gathers over 64 lines", 640 }, { "v3abl: gathers 8 rows x 32 B", 1152 },
                    { "v3abl: NO TESTS, gathers within 1 line", 392 }, { "v3abl: NO TESTS, gathers 8 rows x 32 B", 1160 }, { "v3abl: NO TESTS, gathers 8 rows x 8 B", 136 },
                    { "v3abl: NO TESTS, gathers over 64 lines", 648 } };
                for (auto& e : abl) if (e.v != 40) time_variant<0>(e.nm, pd, darena, dk, n, dd, false, 0, 0, 100 + e.v);
            }
            if (!interior) {
                time_variant<0>("v3x: 2x2-tiled gather addresses (timing only)", pd, darena, dk, n, dd, false, 0, 0, 100 + 32768);
                time_variant<0>("v3x: NO TESTS, 2x2-tiled gather addresses", pd, darena, dk, n, dd, false, 0, 0, 100 + 32776);
                time_variant<0>("v3x: NO TESTS, real gathers", pd, darena, dk, n, dd, false, 0, 0, 100 + 8);
                time_variant<0>("v3x: prio 3 during the fill (bit-exact)", pd, darena, dk, n, dd, false, 0, 0, 100 + 8192);
                time_variant<0>("v3x: prio 3 from the copies on (bit-exact)", pd, darena, dk, n, dd, false, 0, 0, 100 + 16384);
                time_variant<0>("v3x: copies by ds_write_addtid (bit-exact)", pd, darena, dk, n, dd, false, 0, 0, 100 + 4096);
                time_variant<0>("v3x: all rows 8-byte aligned b64 (timing only)", pd, darena, dk, n, dd, false, 0, 0, 100 + 2048);
                time_variant<0>("v3x: b64 + no gather loads", pd, darena, dk, n, dd, false, 0, 0, 100 + 2052);
                time_variant<0>("v3x: b64 + addtid", pd, darena, dk, n, dd, false, 0, 0, 100 + 6144);
                time_variant<0>("v3x: b64 + addtid + no gather loads", pd, darena, dk, n, dd, false, 0, 0, 100 + 6148);
            }
            stamp_report(pd, darena, dk, n, dd, 1);
            if (interior) { time_variant<0>("v3 noclamp", pd, darena, dk, n, dd, false, 0, 0, 2); stamp_report(pd, darena, dk, n, dd, 2); }
            if (!interior) for (int g : { 3072, 2858, 2560, 2500, 2048, 3334, 4096, 6144 })
            { char nm[64]; snprintf(nm, sizeof nm, "persistent grid %d (%.2f kp/wave)", g, (double)n / g); time_variant<0>(nm, pd, darena, dk, n, dd, false, 0, g); }
            if (!interior) {
                // occupancy curve: extra dynamic LDS per workgroup lowers the waves resident per CU (160 KB / (12688 + x))
                for (unsigned x : { 600u, 1800u, 3300u, 5000u, 7300u, 10000u, 14000u })
                { char nm[64]; snprintf(nm, sizeof nm, "production +%u B LDS (%d waves/CU)", x, (int)(163840 / (12688 + x))); time_variant<0>(nm, pd, darena, dk, n, dd, true, x); }
            }
            time_variant<0>("lab, no switch", pd, darena, dk, n, dd, false);
            time_variant<F_COPYUNROLL>("copy unrolled (bit-exact)", pd, darena, dk, n, dd, false);
            time_variant<F_NOCONF>("abl: conflict-free tests", pd, darena, dk, n, dd, false);
            time_variant<F_B64ALL>("abl: all b64 aligned", pd, darena, dk, n, dd, false);
            time_variant<F_B64HALF>("abl: half b64 aligned", pd, darena, dk, n, dd, false);
            time_variant<F_B64ALL | F_NOCONF>("abl: b64 + conflict-free", pd, darena, dk, n, dd, false);
            time_variant<F_NOOUT>("abl: no un-permute", pd, darena, dk, n, dd, false);
            time_variant<F_NOSINCOS>("abl: no sincos", pd, darena, dk, n, dd, false);
            time_variant<F_RPI>("abl?: v_cvt_rpi", pd, darena, dk, n, dd, false);
            time_variant<F_NOFILL>("abl: no gather loads", pd, darena, dk, n, dd, false);
            time_variant<F_NOTEST>("abl: no tests", pd, darena, dk, n, dd, false);
            if (interior) {
                time_variant<F_NOCLAMP>("abl: no clamp", pd, darena, dk, n, dd, false);
                time_variant<F_NOCLAMP | F_RPI>("abl: no clamp + rpi", pd, darena, dk, n, dd, false);
                time_variant<F_NOCLAMP | F_RPI | F_NOOUT | F_COPYUNROLL | F_B64HALF>("abl: noclamp+rpi+noout+cu+b64half", pd, darena, dk, n, dd, false);
                time_variant<F_NOCLAMP | F_RPI | F_NOOUT | F_COPYUNROLL | F_B64ALL | F_NOCONF>("abl: everything", pd, darena, dk, n, dd, false);
            }
        }
    }
    return 0;
}
