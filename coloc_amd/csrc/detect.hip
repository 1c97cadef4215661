// detect.hip -- FAST-9 corner detector + 3x3 non-max suppression + intensity-centroid orientation,
// fully on the GPU, for every pyramid level in one pass.
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
//     214 px wide).  fast_walk_kernel replays the walk for those levels so the output is identical
//     to the compiled reference (tests compare against oracle/_ref).
//   * orientation = fastAtan2(sum r*I, sum c*I) over the 37-pixel disc, 7th-order odd polynomial,
//     fp32 in source order, no FMA contraction.
//
// The reference runs this on the CPU after copying every level back to the host and synchronising
// 8 times per frame (GPUDetector.hpp:262-277); here keypoints never leave HBM: score map -> row
// counts -> exclusive scan -> ordered emit (+ angle), and CLATCH reads the count from device memory.
#include "clc_internal.h"

namespace clc {

struct DetectArgs {
    PyramidDesc pd;
    uint32_t row_begin[CLC_MAX_LEVELS + 1];   // first global row index of each level
    uint32_t threshold;
    uint32_t maxkp;
};

__constant__ const int k_ring_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
__constant__ const int k_ring_dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

__device__ __forceinline__ int level_of_block(const DetectArgs& a, uint32_t blk)
{
    int lv = 0;
#pragma unroll
    for (int i = 1; i < CLC_MAX_LEVELS; ++i)
        if (i < a.pd.levels && blk >= a.pd.blk_begin[i]) lv = i;
    return lv;
}

__device__ __forceinline__ bool fast_pretest(const uint8_t* __restrict__ p, int pitch, int hi, int lo)
{
    const int p9 = p[3 * pitch], p5 = p[3], p1 = p[-3 * pitch], p13 = p[-3];
    const bool b = ((p9 > hi) & (p5 > hi)) | ((p5 > hi) & (p1 > hi)) | ((p1 > hi) & (p13 > hi)) | ((p13 > hi) & (p9 > hi));
    const bool d = ((p9 < lo) & (p5 < lo)) | ((p5 < lo) & (p1 < lo)) | ((p1 < lo) & (p13 < lo)) | ((p13 < lo) & (p9 < lo));
    return b | d;
}

// blk_begin here is rebuilt for the detector: workgroups of 256 pixels over pitch*h bytes of a level
__global__ __launch_bounds__(256) void fast_score_kernel(const DetectArgs a, const uint8_t* __restrict__ arena,
                                                         uint8_t* __restrict__ score)
{
    const int lv = level_of_block(a, blockIdx.x);
    const LevelDesc L = a.pd.lv[lv];
    const uint32_t idx = (blockIdx.x - a.pd.blk_begin[lv]) * 256u + threadIdx.x;
    if (idx >= L.pitch * L.h) return;
    const int y = (int)(idx / L.pitch), x = (int)(idx - (uint32_t)y * L.pitch);
    uint8_t out = 0;
    const int cols = (int)L.w, rows = (int)L.h, pitch = (int)L.pitch;
    if (x >= 3 && x < cols - 3 && y >= 3 && y < rows - 3) {
        const uint8_t* __restrict__ p = arena + L.offset + (size_t)y * pitch + x;
        const int c = *p;
        const int t = (int)a.threshold;
        const int hi = min(c + t, 255), lo = max(c - t, 0);
        if (fast_pretest(p, pitch, hi, lo)) {
            int ring[16];
            uint32_t bm = 0, dm = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                ring[k] = p[k_ring_dy[k] * pitch + k_ring_dx[k]];
                bm |= (ring[k] > hi ? 1u : 0u) << k;
                dm |= (ring[k] < lo ? 1u : 0u) << k;
            }
            // >= 9 contiguous set bits on the 16-cycle
            uint32_t xb = bm | (bm << 16), xd = dm | (dm << 16);
            uint32_t rb = xb & (xb >> 1); rb &= rb >> 2; rb &= rb >> 4; rb &= xb >> 8;
            uint32_t rd = xd & (xd >> 1); rd &= rd >> 2; rd &= rd >> 4; rd &= xd >> 8;
            if (((rb | rd) & 0xFFFFu) != 0u) {
                // corner score: max over the 16 arcs of 9 of max(min(c - ring), -max(c - ring))
                int v[24];
#pragma unroll
                for (int k = 0; k < 24; ++k) v[k] = c - ring[k & 15];
                int best = -32768;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    int mn = v[s], mx = v[s];
#pragma unroll
                    for (int k = 1; k < 9; ++k) { mn = min(mn, v[s + k]); mx = max(mx, v[s + k]); }
                    best = max(best, max(mn, -mx));
                }
                out = (uint8_t)best;
            }
        }
    }
    score[L.offset + idx] = out;
}

// One wave per row of a level whose width is 6 (mod 16): replay the reference's 32-column walk
// (KFAST.h:447-455, 259-265) and drop the last 32 columns when it lands on cols-35.
__global__ __launch_bounds__(64) void fast_walk_kernel(const DetectArgs a, const int lv, const uint8_t* __restrict__ arena,
                                                       uint8_t* __restrict__ score)
{
    const LevelDesc L = a.pd.lv[lv];
    const int cols = (int)L.w, rows = (int)L.h, pitch = (int)L.pitch;
    const int y = 3 + (int)blockIdx.x;
    if (y >= rows - 3) return;
    const int lane = (int)threadIdx.x;
    const uint8_t* __restrict__ row = arena + L.offset + (size_t)y * pitch;
    const int t = (int)a.threshold;
    int j = 3;
    while (j < cols - 35) {
        bool pre = false;
        if (lane < 32) {
            const uint8_t* p = row + j + lane;
            const int c = *p;
            pre = fast_pretest(p, pitch, min(c + t, 255), max(c - t, 0));
        }
        const uint32_t m = (uint32_t)__ballot(pre);
        if (m != 0u && (m & 0xFFFFu) == 0u) j += 16; else j += 32;
    }
    if (j == cols - 35) {
        uint8_t* s = score + L.offset + (size_t)y * pitch;
        if (lane < 32) s[cols - 35 + lane] = 0;
    }
}

__device__ __forceinline__ bool is_keypoint(const uint8_t* __restrict__ s, int pitch, int x, int y, int cols, int rows)
{
    if (x < 3 || x >= cols - 3 || y < 3 || y >= rows - 3) return false;
    const uint8_t* p = s + (size_t)y * pitch + x;
    const int sc = *p;
    if (sc == 0) return false;
    return (sc > p[-1]) & (sc > p[1]) & (sc > p[-pitch - 1]) & (sc > p[-pitch]) & (sc > p[-pitch + 1]) &
           (sc > p[pitch - 1]) & (sc > p[pitch]) & (sc > p[pitch + 1]);
}

__device__ __forceinline__ int level_of_row(const DetectArgs& a, uint32_t grow)
{
    int lv = 0;
#pragma unroll
    for (int i = 1; i < CLC_MAX_LEVELS; ++i)
        if (i < a.pd.levels && grow >= a.row_begin[i]) lv = i;
    return lv;
}

// pass A: keypoints per row (one wave per row, rows of all levels concatenated)
__global__ __launch_bounds__(64) void nms_count_kernel(const DetectArgs a, const uint8_t* __restrict__ score,
                                                       uint32_t* __restrict__ row_count)
{
    const uint32_t grow = blockIdx.x;
    const int lv = level_of_row(a, grow);
    const LevelDesc L = a.pd.lv[lv];
    const int y = (int)(grow - a.row_begin[lv]);
    const uint8_t* __restrict__ s = score + L.offset;
    uint32_t cnt = 0;
    for (int x0 = 0; x0 < (int)L.w; x0 += 64) {
        const bool k = is_keypoint(s, (int)L.pitch, x0 + (int)threadIdx.x, y, (int)L.w, (int)L.h);
        cnt += (uint32_t)__popcll(__ballot(k));
    }
    if (threadIdx.x == 0) row_count[grow] = cnt;
}

// pass B: exclusive scan over the rows (single workgroup; <= a few thousand rows), total -> count[0..1]
__global__ __launch_bounds__(1024) void row_scan_kernel(const uint32_t nrows, const uint32_t* __restrict__ row_count,
                                                        uint32_t* __restrict__ row_off, uint32_t* __restrict__ count,
                                                        const uint32_t maxkp)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (nrows + 1023u) / 1024u;
    const uint32_t r0 = threadIdx.x * per, r1 = min(r0 + per, nrows);
    uint32_t s = 0;
    for (uint32_t r = r0; r < r1; ++r) s += row_count[r];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t st = 1; st < 1024u; st <<= 1) {
        const uint32_t v = threadIdx.x >= st ? part[threadIdx.x - st] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t r = r0; r < r1; ++r) { row_off[r] = run; run += row_count[r]; }
    if (threadIdx.x == 1023u) {
        const uint32_t total = part[1023];
        count[0] = min(total, maxkp);   // keypoints written (level-major order, truncated at capacity)
        count[1] = total;               // keypoints found
    }
}

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

// pass C: ordered emit + orientation (FeatureAngle.h:179-246: rows of 3,5,7,7,7,5,3 pixels)
__global__ __launch_bounds__(64) void nms_emit_kernel(const DetectArgs a, const uint8_t* __restrict__ arena,
                                                      const uint8_t* __restrict__ score,
                                                      const uint32_t* __restrict__ row_off, clc_keypoint* __restrict__ kps)
{
    const uint32_t grow = blockIdx.x;
    const int lv = level_of_row(a, grow);
    const LevelDesc L = a.pd.lv[lv];
    const int y = (int)(grow - a.row_begin[lv]);
    const uint8_t* __restrict__ s = score + L.offset;
    const uint8_t* __restrict__ img = arena + L.offset;
    uint32_t base = row_off[grow];
    for (int x0 = 0; x0 < (int)L.w; x0 += 64) {
        const int x = x0 + (int)threadIdx.x;
        const bool k = is_keypoint(s, (int)L.pitch, x, y, (int)L.w, (int)L.h);
        const uint64_t m = __ballot(k);
        if (k) {
            const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << threadIdx.x) - 1ull));
            if (slot < a.maxkp) {
                int xs = 0, ys = 0;
#pragma unroll
                for (int r = -3; r <= 3; ++r) {
                    const int hw = (r == -3 || r == 3) ? 1 : ((r == -2 || r == 2) ? 2 : 3);
                    const uint8_t* q = img + (size_t)(y + r) * L.pitch + x;
#pragma unroll
                    for (int c = -3; c <= 3; ++c) {
                        if (c < -hw || c > hw) continue;
                        const int v = q[c];
                        xs += c * v;
                        ys += r * v;
                    }
                }
                clc_keypoint kp;
                kp.x = x; kp.y = y;
                kp.score = s[(size_t)y * L.pitch + x];
                kp.angle = fast_atan2((float)(int16_t)ys, (float)(int16_t)xs);
                kp.scale = (uint8_t)lv;
                kps[slot] = kp;
            }
        }
        base += (uint32_t)__popcll(m);
    }
}

hipError_t launch_detect(const PyramidDesc& pd, const uint8_t* arena, uint8_t* score, uint32_t threshold,
                         uint32_t maxkp, uint32_t* d_row_count, uint32_t* d_row_off, uint32_t* d_count,
                         clc_keypoint* d_kps, hipStream_t stream, Profiler* prof)
{
    DetectArgs a;
    a.pd = pd;
    a.threshold = threshold & 0xFFu;
    a.maxkp = maxkp;
    uint32_t blk = 0, rows = 0;
    for (int i = 0; i < pd.levels; ++i) {
        a.pd.blk_begin[i] = blk;
        blk += (pd.lv[i].pitch * pd.lv[i].h + 255u) / 256u;
        a.row_begin[i] = rows;
        rows += pd.lv[i].h;
    }
    for (int i = pd.levels; i <= CLC_MAX_LEVELS; ++i) { a.pd.blk_begin[i] = blk; a.row_begin[i] = rows; }
    prof_mark(prof, CLC_KERNEL_DETECT, true, stream);
    hipLaunchKernelGGL(fast_score_kernel, dim3(blk), dim3(256), 0, stream, a, arena, score);
    for (int i = 0; i < pd.levels; ++i)
        if (pd.lv[i].w % 16u == 6u && pd.lv[i].h > 6u)
            hipLaunchKernelGGL(fast_walk_kernel, dim3(pd.lv[i].h - 6u), dim3(64), 0, stream, a, i, arena, score);
    hipLaunchKernelGGL(nms_count_kernel, dim3(rows), dim3(64), 0, stream, a, (const uint8_t*)score, d_row_count);
    hipLaunchKernelGGL(row_scan_kernel, dim3(1), dim3(1024), 0, stream, rows, (const uint32_t*)d_row_count, d_row_off,
                       d_count, maxkp);
    hipLaunchKernelGGL(nms_emit_kernel, dim3(rows), dim3(64), 0, stream, a, arena, (const uint8_t*)score,
                       (const uint32_t*)d_row_off, d_kps);
    prof_mark(prof, CLC_KERNEL_DETECT, false, stream);
    return hipGetLastError();
}

uint32_t detect_total_rows(const PyramidDesc& pd)
{
    uint32_t rows = 0;
    for (int i = 0; i < pd.levels; ++i) rows += pd.lv[i].h;
    return rows;
}

} // namespace clc
