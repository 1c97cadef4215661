// detect.hip -- FAST-9 corner detector + 3x3 non-max suppression + intensity-centroid orientation,
// fully on the GPU, for every pyramid level of up to CLC_MAX_BATCH cameras in TWO launches
// (SURVEY.md 8 row a-8 / f-1).
//
// Semantics: reference include/coloc/KFAST.h:164-500 (`KFAST<true,true>`) and
// include/coloc/FeatureAngle.h:160-246 as called from GPUDetector::detectAndDescribe
// (include/coloc/GPUDetector.hpp:262-277):
//   * FAST-9: >= 9 contiguous of the 16 Bresenham-ring pixels brighter than sat(p+t) or darker than
//     sat(p-t) (KFAST.h:178-184, 272-290); score = max over the 16 arcs of 9 of
//     max(min(p-ring), -max(p-ring)) (:300-374); strict 3x3 non-max suppression on the score map
//     (:492); candidates in rows [3, rows-3), cols [3, cols-3); keypoint order = (level, y, x)
//     ascending, which is what the reference's row bands concatenate to.
//   * the reference walks a row in 32-column blocks with a 16-column retreat (:259-265) and masks the
//     last partial block with (1 << (cols-j-3)) - 1 (:245); when the walk lands exactly on
//     j == cols-35 the shift count is 32 = a shift by 0 on x86, so the last 32 columns of THAT ROW
//     are dropped.  Only widths with cols % 16 == 6 can get there (level 6 of a 640-wide pyramid is
//     214 px wide).  The tiles of such a level that touch its last 35 columns replay the walk for
//     their own rows (one flag per 16-column group of the row -- all the walk ever asks --, then
//     one lane per row walks them) and clear the dropped columns' scores in front of the suppression,
//     so the output is identical to the compiled reference (tests compare against oracle/_ref).
//   * orientation = fastAtan2(sum r*I, sum c*I) over the 37-pixel disc, 7th-order odd polynomial,
//     fp32 in source order, no FMA contraction.
//
// The reference runs this on the CPU after copying every level back to the host and synchronising
// 8 times per frame (GPUDetector.hpp:262-277); here keypoints never leave HBM.
//
// MI355X shape (round 4; round 1's five dependent launches -- score / walk / count / scan / emit, one pixel per thread,
// one wave per row -- took 40 us of kernels + 4 boundaries for a 640 x 480 frame):
//   launch A  detect_tile_kernel: one 256-thread workgroup per 64 x 16 tile of a level (all levels, all cameras: ~1000
//             tiles per 640 x 480 frame).  The tile + 4 px halo is staged into LDS with dword loads; the cardinal
//             pre-test runs densely over the tile + 1 px border and COMPACTS the survivors (a few % of the pixels) into an
//             LDS list; the corner score then runs on that list only (the 9-of-16 segment test IS "score > threshold", so
//             no ring bitmask is built -- round 5), so the expensive path is paid per candidate, not per wave that holds
//             one; strict 3 x 3 non-max suppression reads the LDS score tile.  Out: one 64-bit keypoint mask per tile row,
//             the tile's keypoint count, and the score byte of each keypoint.
//   launch B  detect_emit_kernel: one workgroup per 16-row band of a level.  It sums the counts of the tiles in front of
//             its band (level-major order; <= 13 KB, L2 resident -- no scan launch, no atomics, nothing to re-arm), scans
//             the band's mask words in (y, x) order and writes the keypoints, orientation included, at their final slots.
//   (Round 5 built the two as ONE launch -- the last tile of a band to arrive emits the band, prefix from per-band arrival words --
//   and measured it at 24.6 us per frame against 13.3 for the two launches: every step of publish -> arrive -> poll crosses the XCDs
//   at 1.5-7 us under the load of 1 000 workgroups doing the same; profiles/r05_detect_notes.txt, r05_detect_fused_emit.patch.)
#include "clc_internal.h"

namespace clc {

static constexpr int kTileW = 64, kTileH = 16;
static constexpr int kImgRows = kTileH + 8, kImgStride = kTileW + 8;        // tile + 3 px ring + 1 px NMS border, each side
static constexpr int kScRows = kTileH + 2, kScStride = kTileW + 8;          // score tile: column c of the region at byte c + 3
static constexpr int kRegion = (kTileW + 2) * (kTileH + 2);                 // pixels whose score the tile needs
static constexpr int kCandPerWave = 64 * ((kRegion + 255) / 256);           // a wave pre-tests every fourth 64-pixel run of the region
static constexpr int kWalkWords = 2 * ((CLC_DETECT_MAX_WIDTH + 255) / 256) + 2; // a row's 16-column group flags, a nibble each (+ padding)

struct DetectArgs {
    PyramidDesc pd;
    uint32_t tile_begin[CLC_MAX_LEVELS + 1];  // first tile of each level (tiles in level, ty, tx order)
    uint32_t band_begin[CLC_MAX_LEVELS + 1];  // first 16-row band of each level
    uint32_t tiles_x[CLC_MAX_LEVELS];
    uint32_t n_tiles, rot;                    // the launch starts at tile `rot` (the first level that replays the walk: its tiles take longest)
    uint32_t n_bands;                         // 16-row bands of all levels
    uint32_t threshold, maxkp;
    uint32_t slot_stride;                     // bytes between the pyramids / score maps of consecutive cameras
    clc_keypoint* kps[kMaxBatch];
    uint32_t* count[kMaxBatch];               // {written, found}
};

__device__ __forceinline__ int level_of(const uint32_t* begin, int levels, uint32_t idx)
{
    int lv = 0;
#pragma unroll
    for (int i = 1; i < CLC_MAX_LEVELS; ++i)
        if (i < levels && idx >= begin[i]) lv = i;
    return lv;
}

// cardinal pre-test (KFAST.h:230-244): two ADJACENT of the four compass ring pixels beyond the threshold on the same side.  Every adjacent
// pair takes one of {p1, p9} and one of {p5, p13}, so "some adjacent pair brighter than c + t" is min(max(p1, p9), max(p5, p13)) > c + t
// and the dark side mirrors it; the saturations of the reference (min(c + t, 255), max(c - t, 0)) change nothing for bytes
// (p > min(c + t, 255) <=> p > c + t when p <= 255).  Ten integer instructions where the eight compares + fourteen mask operations took 30.
__device__ __forceinline__ bool pretest4(int c, int p1, int p5, int p9, int p13, int t)
{
    const int bright = min(max(p1, p9), max(p5, p13)), dark = max(min(p1, p9), min(p5, p13));
    return max(bright - c, c - dark) > t;
}

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }   // v_min3_i32 / v_max3_i32
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// ring offsets inside the LDS image tile, relative to the top-left of the pixel's 7 x 7 neighbourhood (all >= 0: immediates)
#define RING_OFF(k) ((3 + k_dy[k]) * kImgStride + 3 + k_dx[k])

// include/coloc/FeatureAngle.h:160-177
__device__ __forceinline__ float fast_atan2(const float y, const float x)
{
    const float PI = 3.1415927f;
    const float ax = fabsf(x), ay = fabsf(y);
    float r;
    if (ax >= ay) {
        const float c = ay / (ax + 1.17549435e-38f);
        const float cc = c * c;
        r = (((-0.0443265555479f * cc + 0.1555786518f) * cc - 0.325808397f) * cc + 0.9997878412f) * c;
    } else {
        const float c = ax / (ay + 1.17549435e-38f);
        const float cc = c * c;
        r = PI * 0.5f - (((-0.0443265555479f * cc + 0.1555786518f) * cc - 0.325808397f) * cc + 0.9997878412f) * c;
    }
    if (x < 0.0f) r = PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, s);
        if (lane >= (uint32_t)s) v += o;
    }
    return v;
}

static constexpr uint32_t kEmitList = 2048u;              // keypoints listed per pass (a 640-wide band holds ~35)

// The keypoints of one 16-row band, in (y, x) order, to their final slots carry, carry + 1, ..: scan of the band's mask words (row-major over
// the band's tiles), orientation per keypoint (FeatureAngle.h:179-246: rows of 3,5,7,7,7,5,3 pixels).  All 256 threads of a workgroup.
// first_word: the caller's early load of this thread's first mask word (its latency then overlaps whatever the caller did in between).
__device__ __forceinline__ uint32_t emit_band(const DetectArgs& a, const LevelDesc& L, int lv, uint32_t ty, uint32_t ntx, uint32_t carry,
                                          const uint64_t* __restrict__ mask /* of the band's first tile */, uint64_t first_word,
                                          const uint8_t* __restrict__ img, const uint8_t* __restrict__ score, clc_keypoint* __restrict__ kps,
                                          uint32_t* s_part, uint16_t* s_list)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nwords = (uint32_t)kTileH * ntx;
    for (uint32_t w0 = 0; w0 < nwords; w0 += 256u) {
        const uint32_t i = w0 + tid;
        const uint32_t r = i / ntx, tx = i - r * ntx;
        const uint64_t m = w0 == 0u ? first_word : (i < nwords ? mask[(size_t)tx * kTileH + r] : 0ull);
        const uint32_t c = (uint32_t)__popcll(m);
        const uint32_t inc = wave_inclusive_scan(c, lane);
        if (lane == 63u) s_part[wave] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += s_part[w];
        // (strict 3 x 3 suppression leaves at most 32 keypoints per 64-bit word: 8 192 per 256 words at most)
        const uint32_t total = min(s_part[0] + s_part[1] + s_part[2] + s_part[3], 256u * 32u);
        // the keypoints of these 256 words, in order, into one list: afterwards EVERY thread takes one keypoint (a thread that walked
        // the bits of its own word paid one load latency per keypoint of that word); kEmitList at a time
        for (uint32_t base = 0; base < total; base += kEmitList) {
            uint32_t k = before + inc - c - base;             // (wraps below `base`: then not in this pass)
            uint64_t mm = m;
            while (mm != 0ull) {
                const uint32_t b = (uint32_t)__builtin_ctzll(mm);
                mm &= mm - 1ull;
                if (k < kEmitList) s_list[k] = (uint16_t)((tx * kTileW + b) | (r << 12));
                ++k;
            }
            __syncthreads();
            const uint32_t n_here = min(total - base, kEmitList);
            for (uint32_t q = tid; q < n_here; q += 256u) {
                const uint32_t slot = carry + base + q;
                if (slot >= a.maxkp) break;
                const uint32_t e = s_list[q];
                // keypoints lie in [3, w - 4] x [3, h - 4] (KFAST.h:431,455): the clamp is the identity on them -- it only keeps a mask that did
                // not come from detect_tile_kernel from reading outside the level
                const int x = min(max((int)(e & 0xFFFu), 3), (int)L.w - 4), y = min(max((int)(ty * kTileH + (e >> 12)), 3), (int)L.h - 4);
                int xs = 0, ys = 0;
#pragma unroll
                for (int rr = -3; rr <= 3; ++rr) {
                    const int hw = (rr == -3 || rr == 3) ? 1 : ((rr == -2 || rr == 2) ? 2 : 3);
                    const uint8_t* q8 = img + (size_t)(y + rr) * L.pitch + x;
#pragma unroll
                    for (int cc = -3; cc <= 3; ++cc) {
                        if (cc < -hw || cc > hw) continue;
                        const int v = q8[cc];
                        xs += cc * v;
                        ys += rr * v;
                    }
                }
                clc_keypoint kp;
                kp.x = x; kp.y = y;
                kp.score = score[(size_t)y * L.pitch + x];
                kp.angle = fast_atan2((float)(int16_t)ys, (float)(int16_t)xs);
                kp.scale = (uint8_t)lv;
                kps[slot] = kp;
            }
            __syncthreads();
        }
        carry += total;
    }
    return carry;                                             // = the keypoints in front of the next band (the same in every thread)
}

__global__ __launch_bounds__(256) void detect_tile_kernel(const DetectArgs a, const uint8_t* __restrict__ arena_base,
                                                          uint8_t* __restrict__ score_base, uint64_t* __restrict__ mask_base,
                                                          uint32_t* __restrict__ tcount_base)
{
    constexpr int k_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
    constexpr int k_dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
    __shared__ __attribute__((aligned(16))) uint8_t s_img[kImgRows * kImgStride];
    __shared__ __attribute__((aligned(16))) uint8_t s_sc[kScRows * kScStride];
    __shared__ uint16_t s_cand[4][kCandPerWave];      // wave-private candidate lists: no atomic, no barrier between pre-test and ring stage
    __shared__ uint32_t s_walk[kScRows][kWalkWords];
    __shared__ uint32_t s_mask[kTileH][2];
    __shared__ uint32_t s_drop[kScRows];
    __shared__ uint32_t s_nkp;

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t cam = blockIdx.y;
    uint32_t tile = blockIdx.x + a.rot;
    if (tile >= a.n_tiles) tile -= a.n_tiles;
    const int lv = level_of(a.tile_begin, a.pd.levels, tile);
    const LevelDesc L = a.pd.lv[lv];
    const uint32_t tl = tile - a.tile_begin[lv];
    const int ty = (int)(tl / a.tiles_x[lv]), tx = (int)(tl - (uint32_t)ty * a.tiles_x[lv]);
    const int x0 = tx * kTileW, y0 = ty * kTileH;
    const int cols = (int)L.w, rows = (int)L.h, pitch = (int)L.pitch;
    const int t = (int)a.threshold;
    const uint8_t* __restrict__ img = arena_base + (size_t)cam * a.slot_stride + L.offset;

    // ---- stage the tile (+ 4 px halo) and clear the work areas ---------------------------------
    {   // (both dwords of a thread are requested before either is stored: one load latency, not two)
        constexpr uint32_t kDwords = (uint32_t)(kImgRows * (kImgStride / 4));
        static_assert(kDwords <= 512u, "two staging dwords per thread");
        uint32_t v[2] = { 0u, 0u };
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t i = tid + 256u * (uint32_t)k;
            const int r = (int)(i / (kImgStride / 4)), d = (int)(i - (uint32_t)r * (kImgStride / 4));
            const int gy = y0 - 4 + r, gx = x0 - 4 + 4 * d;
            const bool in = i < kDwords && gy >= 0 && gy < rows && gx >= 0 && gx < pitch;
            const uint32_t w = *reinterpret_cast<const uint32_t*>(img + (size_t)min(max(gy, 0), rows - 1) * pitch + min(max(gx, 0), pitch - 4));
            v[k] = in ? w : 0u;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t i = tid + 256u * (uint32_t)k;
            if (i < kDwords) reinterpret_cast<uint32_t*>(s_img)[i] = v[k];
        }
    }
    for (uint32_t i = tid; i < (uint32_t)(kScRows * kScStride / 4); i += 256u) reinterpret_cast<uint32_t*>(s_sc)[i] = 0u;
    if (tid < (uint32_t)kTileH * 2u) s_mask[tid >> 1][tid & 1u] = 0u;
    if (tid == 0) s_nkp = 0u;

    // ---- KFAST.h:245 replay: does the walk of row gy land on cols - 35?  (levels of width 6 mod 16, tiles at the right end)
    // The answer (s_drop) is only needed when the scores go into the suppression: rows of at most 256 columns -- every level of a 640-wide
    // pyramid that can get here -- issue the replay's global loads HERE and turn them into flags behind the ring stage, so that their
    // latency runs under the tile's own work (round 4 waited for them in front of it: 3.3 us on the launch's longest tiles).
    const bool walk = (cols % 16 == 6) && cols >= 38 && rows >= 7 && (x0 + kTileW + 1 >= cols - 35);   // (rows < 7: no candidate row at all)
    const int nunits = (cols + 255) >> 8;                        // 256-column units per row (<= 16)
    const bool walk_regs = walk && nunits == 1;
    // The walk only ever asks two things of a 32-column block at j = 3 + 16 g: is any pre-test bit set, and is its low half empty --
    // i.e. it needs any(g) = "some pixel of columns 3 + 16 g .. 18 + 16 g passes the pre-test" per 16-column group.  A lane takes four
    // columns 4 l .. 4 l + 3 of a row as dwords (3 of the centre row, 1 each of the rows 3 above and below: five loads instead
    // of twenty byte loads), a wave instruction covers 256 columns, and all the loads of the wave's rows (w, w + 4, ..: five at most)
    // are in flight together.  Lane 4 g' + r: r = 0 -> its columns 0..2 belong to group g' - 1, column 3 to g'; r != 0 -> all four to g'.
    // C = ballot(own group) | ballot(previous group) >> 4 then holds any(g) in nibble g.
    uint32_t wk0[5], wk1[5], wk2[5], wkt[5], wkb[5];
    auto walk_load = [&](int c) {                                // (addresses clamped into the level -- no branch per load; walk_flags only uses
                                                                 //  what lies inside: 3 <= x <= cols - 5 on rows 3 .. rows - 4)
        const int xl = 256 * c + 4 * (int)lane;                  // first of this lane's four columns
        const int xc = min(xl, pitch - 4), xm = max(xc - 4, 0), xp = min(xl + 4, pitch - 4);
#pragma unroll
        for (int ri = 0; ri < 5; ++ri) {
            const int r = (int)wave + 4 * ri;
            const int gy = y0 - 1 + r;
            const uint8_t* p = img + (size_t)min(max(gy, 3), rows - 4) * pitch;
            wk0[ri] = *reinterpret_cast<const uint32_t*>(p + xm);
            wk1[ri] = *reinterpret_cast<const uint32_t*>(p + xc);
            wk2[ri] = *reinterpret_cast<const uint32_t*>(p + xp);
            wkt[ri] = *reinterpret_cast<const uint32_t*>(p - 3 * pitch + xc);
            wkb[ri] = *reinterpret_cast<const uint32_t*>(p + 3 * pitch + xc);
        }
    };
    // group flags of row wave + 4 ri, unit c: nibble g = any(16 c + g); `carry_in` = the unit's columns 0..2 hold a bit of the unit before
    auto walk_flags = [&](int c, int ri, bool& carry_in) -> uint64_t {
        const int xl = 256 * c + 4 * (int)lane;
        const int r = (int)wave + 4 * ri;
        const int gy = y0 - 1 + r;
        const bool ok = r < kScRows && gy >= 3 && gy < rows - 3;
        const uint64_t lo64 = (uint64_t)wk0[ri] | ((uint64_t)wk1[ri] << 32);     // columns xl - 4 .. xl + 3
        const uint64_t hi64 = (uint64_t)wk1[ri] | ((uint64_t)wk2[ri] << 32);     // columns xl .. xl + 7
        bool own = false, prev = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = xl + k;
            const int cc = (int)((wk1[ri] >> (8 * k)) & 0xFFu);
            const int left = (int)((lo64 >> (8 * (k + 1))) & 0xFFu), right = (int)((hi64 >> (8 * (k + 3))) & 0xFFu);
            const int up = (int)((wkt[ri] >> (8 * k)) & 0xFFu), down = (int)((wkb[ri] >> (8 * k)) & 0xFFu);
            const bool pre = ok && x >= 3 && x <= cols - 5 && pretest4(cc, up, right, down, left, t);
            if ((lane & 3u) == 0u && k < 3) prev |= pre; else own |= pre;
        }
        const uint64_t b_own = __ballot(own), b_prev = __ballot(prev);
        carry_in = (b_prev & 1ull) != 0ull;
        return b_own | (b_prev >> 4);
    };
    if (walk_regs) walk_load(0);
    if (!walk_regs) {
        if (tid < (uint32_t)kScRows) s_drop[tid] = 0u;           // (with the replay, every entry is written by the thread that walks its row)
    }
    if (walk && !walk_regs) {
        // wider rows: the flags of a row go through LDS, one thread per row walks them (in front of the tile's work, as in round 4)
        for (int c = 0; c < nunits; ++c) {
            walk_load(c);
#pragma unroll
            for (int ri = 0; ri < 5; ++ri) {
                const int r = (int)wave + 4 * ri;
                bool carry_in;
                const uint64_t cmask = walk_flags(c, ri, carry_in);
                if (lane == 0 && r < kScRows) {
                    s_walk[r][2 * c] = (uint32_t)cmask; s_walk[r][2 * c + 1] = (uint32_t)(cmask >> 32);
                    // columns 0..2 of a unit behind the first belong to the LAST group of the unit before (this lane stored it a step ago)
                    if (c > 0 && carry_in) s_walk[r][2 * c - 1] |= 0xF0000000u;
                }
            }
        }
        if (tid < (uint32_t)kScRows) { s_walk[tid][2 * nunits] = 0u; s_walk[tid][2 * nunits + 1] = 0u; }
        __syncthreads();
        if (tid < (uint32_t)kScRows) {
            const int gy = y0 - 1 + (int)tid;
            uint32_t drop = 0u;
            if (gy >= 3 && gy < rows - 3) {
                int g = 0;                                                          // j = 3 + 16 g
                while (3 + 16 * g < cols - 35) {
                    const uint32_t w0 = s_walk[tid][g >> 3], w1 = s_walk[tid][(g + 1) >> 3];
                    const bool a0 = ((w0 >> (4 * (g & 7))) & 0xFu) != 0u, a1 = ((w1 >> (4 * ((g + 1) & 7))) & 0xFu) != 0u;
                    g += (!a0 && a1) ? 1 : 2;                                       // KFAST.h:259-265: retreat when only the high half has bits
                }
                drop = (3 + 16 * g == cols - 35) ? 1u : 0u;
            }
            s_drop[tid] = drop;
        }
    }
    __syncthreads();

    // ---- phase 1: dense cardinal pre-test over the tile + 1 px border; every wave compacts its survivors into a list of its own ----
    // (columns that the replay drops are scored like the others and cleared in front of the suppression)
    uint32_t ncand = 0;                                        // wave-uniform: candidates of THIS wave
    for (uint32_t i0 = 0; i0 < (uint32_t)kRegion; i0 += 256u) {
        const uint32_t i = i0 + tid;
        const int r = (int)(i / (kTileW + 2)), c = (int)(i - (uint32_t)r * (kTileW + 2));
        const int gx = x0 - 1 + c, gy = y0 - 1 + r;
        bool pre = false;
        if (i < (uint32_t)kRegion && gx >= 3 && gx < cols - 3 && gy >= 3 && gy < rows - 3) {
            const uint8_t* p = s_img + (r + 3) * kImgStride + (c + 3);
            pre = pretest4(p[0], p[-3 * kImgStride], p[3], p[3 * kImgStride], p[-3], t);
        }
        const uint64_t m = __ballot(pre);
        if (pre) s_cand[wave][ncand + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)((r << 8) | c);
        ncand += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // the list is read by the wave that wrote it
    __builtin_amdgcn_wave_barrier();

    // ---- phase 2: corner score, one candidate per lane ------------------------------------------
    // KFAST.h:300-374: score = max over the 16 arcs of 9 of max(min(c - ring), -max(c - ring)).  An arc of nine all brighter than c + t is an
    // arc whose max(c - ring) < -t, one all darker an arc whose min(c - ring) > t (the saturations drop out as in pretest4): the 9-of-16
    // segment test (:178-184, 272-290) IS "score > t", so the score is computed once and no ring bitmask is built.  min / max over nine
    // consecutive = three threes (v_min3 / v_max3), shared between the arcs.
    for (uint32_t i = lane; i < ncand; i += 64u) {
        const uint32_t rc = s_cand[wave][i];
        const int r = (int)(rc >> 8), c = (int)(rc & 0xFFu);
        const uint8_t* q = s_img + r * kImgStride + c;         // top-left of the 7 x 7 neighbourhood
        const int ctr = q[3 * kImgStride + 3];
        int v[16], mn3[16], mx3[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = ctr - (int)q[RING_OFF(k)];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            mn3[k] = min3i(v[k], v[(k + 1) & 15], v[(k + 2) & 15]);
            mx3[k] = max3i(v[k], v[(k + 1) & 15], v[(k + 2) & 15]);
        }
        int dark = -32768, bright = 32767;                     // max over the arcs of min(c - ring), min over the arcs of max(c - ring)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            dark = max(dark, min3i(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]));
            bright = min(bright, max3i(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]));
        }
        const int best = max(dark, -bright);
        if (best > t) s_sc[r * kScStride + c + 3] = (uint8_t)best;
    }

    // ---- the replay's verdict for narrow rows: flags from the loads issued at the top, the walk in registers on the lane that holds its row ----
    if (walk_regs) {
        uint64_t row_flags = 0;                                  // lane ri of a wave holds the group flags of row wave + 4 ri
#pragma unroll
        for (int ri = 0; ri < 5; ++ri) {
            bool carry_in;
            const uint64_t f = walk_flags(0, ri, carry_in);
            if ((int)lane == ri) row_flags = f;
        }
        const int r = (int)wave + 4 * (int)lane;
        const int gy = y0 - 1 + r;
        if (lane < 5u && r < kScRows) {
            uint32_t drop = 0u;
            if (gy >= 3 && gy < rows - 3) {
                int g = 0;
                while (3 + 16 * g < cols - 35) {
                    const bool a0 = ((row_flags >> (4 * g)) & 0xFull) != 0ull, a1 = ((row_flags >> (4 * g + 4)) & 0xFull) != 0ull;
                    g += (!a0 && a1) ? 1 : 2;
                }
                drop = (3 + 16 * g == cols - 35) ? 1u : 0u;
            }
            s_drop[r] = drop;
        }
    }
    __syncthreads();
    if (walk) {
        // KFAST.h:245: a row whose walk lands on cols - 35 loses its last 32 columns (cols - 35 .. cols - 4) BEFORE the suppression
        const int c_first = cols - 35 - (x0 - 1);                // region column of cols - 35
        for (uint32_t i = tid; i < (uint32_t)(kScRows * 32); i += 256u) {
            const int r = (int)(i >> 5), c = c_first + (int)(i & 31u);
            if (s_drop[r] && c >= 0 && c < kTileW + 2) s_sc[r * kScStride + c + 3] = 0u;
        }
        __syncthreads();
    }

    // ---- strict 3 x 3 non-max suppression of the tile's own 64 x 16 pixels (thread = 4 pixels of a row) ----
    {
        const int r = (int)(tid >> 4) + 1, d = (int)(tid & 15u);               // region row 1..16, dword d of the interior
        const uint8_t* srow = s_sc + r * kScStride + 4 + 4 * d;                 // region column c = 1 + 4 d sits at byte c + 3
        const uint32_t four = *reinterpret_cast<const uint32_t*>(srow);
        if (four != 0u) {
            uint8_t* __restrict__ gscore = score_base + (size_t)cam * a.slot_stride + L.offset;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sc = (int)((four >> (8 * k)) & 0xFFu);
                if (sc == 0) continue;
                const uint8_t* p = srow + k;
                const bool kp = (sc > p[-1]) & (sc > p[1]) & (sc > p[-kScStride - 1]) & (sc > p[-kScStride]) & (sc > p[-kScStride + 1]) &
                                (sc > p[kScStride - 1]) & (sc > p[kScStride]) & (sc > p[kScStride + 1]);
                if (kp) {
                    const int xl = 4 * d + k;                                   // column inside the tile
                    atomicOr(&s_mask[r - 1][xl >> 5], 1u << (xl & 31));
                    atomicAdd(&s_nkp, 1u);
                    gscore[(size_t)(y0 + r - 1) * pitch + x0 + xl] = (uint8_t)sc;
                }
            }
        }
    }
    __syncthreads();
    if (tid < (uint32_t)kTileH)
        mask_base[((size_t)cam * a.n_tiles + tile) * kTileH + tid] = (uint64_t)s_mask[tid][0] | ((uint64_t)s_mask[tid][1] << 32);
    if (tid == 0) tcount_base[(size_t)cam * a.n_tiles + tile] = s_nkp;
}

// One workgroup per 16-row band; the keypoints in front of the band = the sum of the tile counts in front of it (level-major order;
// <= 13 KB -- no scan launch, no atomics, nothing to re-arm).
__global__ __launch_bounds__(256) void detect_emit_kernel(const DetectArgs a, const uint8_t* __restrict__ arena_base,
                                                          const uint8_t* __restrict__ score_base, const uint64_t* __restrict__ mask_base,
                                                          const uint32_t* __restrict__ tcount_base)
{
    __shared__ uint32_t s_part[8];
    __shared__ uint16_t s_list[kEmitList];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t cam = blockIdx.y, band = blockIdx.x;
    const int lv = level_of(a.band_begin, a.pd.levels, band);
    const LevelDesc L = a.pd.lv[lv];
    const uint32_t ty = band - a.band_begin[lv], ntx = a.tiles_x[lv];
    const uint32_t t0 = a.tile_begin[lv] + ty * ntx;
    const uint32_t* __restrict__ tcount = tcount_base + (size_t)cam * a.n_tiles;
    const uint64_t* __restrict__ mask = mask_base + ((size_t)cam * a.n_tiles + t0) * kTileH;

    // this thread's first mask word (loaded before the reduction so that both latencies overlap)
    const uint64_t first_word = tid < (uint32_t)kTileH * ntx ? mask[(size_t)(tid % ntx) * kTileH + tid / ntx] : 0ull;
    uint32_t s = 0;
    for (uint32_t i = tid; i < t0; i += 256u) s += tcount[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += (uint32_t)__shfl_xor((int)s, o);
    if (lane == 0) s_part[4 + wave] = s;
    __syncthreads();
    const uint32_t carry = s_part[4] + s_part[5] + s_part[6] + s_part[7];
    const uint32_t end = emit_band(a, L, lv, ty, ntx, carry, mask, first_word, arena_base + (size_t)cam * a.slot_stride + L.offset,
                                   score_base + (size_t)cam * a.slot_stride + L.offset, a.kps[cam], s_part, s_list);
    // the last band of the camera knows the total
    if (tid == 0 && band + 1 == a.n_bands) {
        a.count[cam][0] = min(end, a.maxkp);   // keypoints written (level-major order, truncated at capacity)
        a.count[cam][1] = end;                 // keypoints found
    }
}

uint32_t detect_total_tiles(const PyramidDesc& pd)
{
    uint32_t n = 0;
    for (int i = 0; i < pd.levels; ++i) n += ((pd.lv[i].w + kTileW - 1) / kTileW) * ((pd.lv[i].h + kTileH - 1) / kTileH);
    return n;
}

hipError_t launch_detect(const PyramidDesc& pd, const uint8_t* arena, size_t slot_stride, int n_img, uint8_t* score, uint64_t* d_mask,
                         uint32_t* d_tcount, uint32_t threshold, uint32_t maxkp, clc_keypoint* const* d_kps, uint32_t* const* d_count,
                         hipStream_t stream, Profiler* prof)
{
    if (n_img <= 0) return hipSuccess;
    if (n_img > kMaxBatch || slot_stride > 0xFFFFFFFFull) return hipErrorInvalidValue;
    DetectArgs a;
    a.pd = pd;
    a.threshold = threshold & 0xFFu;
    a.maxkp = maxkp;
    a.slot_stride = (uint32_t)slot_stride;
    uint32_t tiles = 0, bands = 0;
    a.rot = 0;
    bool rot_set = false;
    for (int i = 0; i < pd.levels; ++i) {
        if (pd.lv[i].w > CLC_DETECT_MAX_WIDTH) return hipErrorInvalidValue;
        a.tile_begin[i] = tiles;
        a.band_begin[i] = bands;
        a.tiles_x[i] = (pd.lv[i].w + kTileW - 1) / kTileW;
        const uint32_t tiles_y = (pd.lv[i].h + kTileH - 1) / kTileH;
        if (!rot_set && pd.lv[i].w % 16u == 6u && pd.lv[i].w >= 38u) { a.rot = tiles; rot_set = true; }
        tiles += a.tiles_x[i] * tiles_y;
        bands += tiles_y;
    }
    for (int i = pd.levels; i <= CLC_MAX_LEVELS; ++i) { a.tile_begin[i] = tiles; a.band_begin[i] = bands; if (i < CLC_MAX_LEVELS) a.tiles_x[i] = 1; }
    a.n_tiles = tiles;
    a.n_bands = bands;
    for (int b = 0; b < kMaxBatch; ++b) { a.kps[b] = b < n_img ? d_kps[b] : nullptr; a.count[b] = b < n_img ? d_count[b] : nullptr; }
    if (tiles == 0) return hipSuccess;
    prof_mark(prof, CLC_KERNEL_DETECT, true, stream);
    hipLaunchKernelGGL(detect_tile_kernel, dim3(tiles, (uint32_t)n_img), dim3(256), 0, stream, a, arena, score, d_mask, d_tcount);
    hipLaunchKernelGGL(detect_emit_kernel, dim3(bands, (uint32_t)n_img), dim3(256), 0, stream, a, arena, (const uint8_t*)score,
                       (const uint64_t*)d_mask, (const uint32_t*)d_tcount);
    prof_mark(prof, CLC_KERNEL_DETECT, false, stream);
    return hipGetLastError();
}

} // namespace clc

