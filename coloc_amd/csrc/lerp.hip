// lerp.hip -- scale-space pyramid: bilinear resample of level 0 into levels 1..L-1 (u8 -> u8).
//
// Semantics: reference src/CUDALERP.cu:157-178 for one output pixel and
// include/coloc/GPUDetector.hpp:109-114,249-254 for level sizes / factors (every level is
// resampled FROM LEVEL 0 with gxs = gys = f_i, f_i = f_{i-1} * scale_factor in fp32).
// Texture-unit behaviour restated in plain arithmetic: tex2Dgather footprint = the four texels
// (i,j),(i+1,j),(i,j+1),(i+1,j+1) with i = floor(fx), j = floor(fy), clamp-to-edge addressing
// (GPUDetector.hpp:96-97) and normalized-float reads u8/255.0f (:239).  fp32, evaluation order as
// written in the source, no FMA contraction (the file is built with -ffp-contract=off).
//
// MI355X shape: the reference issues 7 launches on 7 streams (GPUDetector.hpp:250-255); here ALL
// levels AND the level-0 copy into the pyramid arena are one launch -- for the frames of up to CLC_MAX_BATCH cameras at once (blockIdx.y = camera).  Each lane produces 4 horizontally adjacent output pixels and stores one
// dword, so a wave writes 256 contiguous bytes; the source taps are plain byte loads from the
// L2-resident level-0 image (640x480 = 300 KB).  HBM-trivial: writes 2.09 x W x H bytes.
#include "clc_internal.h"

namespace clc {

struct LerpArgs {
    PyramidDesc pd;
    uint32_t src_pitch;      // bytes per row of the caller's level-0 image
    uint32_t copy_blocks;    // leading workgroups that copy level 0 into the arena (0 = already there)
    uint32_t slot_stride;    // bytes between the pyramids of consecutive images (blockIdx.y = image)
    const uint8_t* src[kMaxBatch];   // level-0 image of each camera
};

// normalized-float texel read: u8 / 255.0f (GPUDetector.hpp:239).  The correctly rounded quotient is taken from a
// 256-entry table that each workgroup fills with that very division (one per thread) -- a tap then costs one LDS
// read instead of a ~10-instruction IEEE division, 16 times per lane.
__device__ __forceinline__ float tap(const uint8_t* __restrict__ img, uint32_t pitch, int W, int H, int x, int y,
                                     const float* __restrict__ unorm)
{
    x = min(max(x, 0), W - 1);
    y = min(max(y, 0), H - 1);
    return unorm[img[(size_t)y * pitch + (size_t)x]];
}

__global__ __launch_bounds__(256) void pyramid_kernel(const LerpArgs a, uint8_t* __restrict__ arena_base)
{
    uint8_t* __restrict__ arena = arena_base + (size_t)blockIdx.y * a.slot_stride;
    const uint8_t* __restrict__ src0 = a.src[blockIdx.y];
    const LevelDesc L0 = a.pd.lv[0];
    if (blockIdx.x < a.copy_blocks) {
        // level 0: copy the caller's image into the arena (dword stores; CLATCH samples it from there).
        // The resample workgroups below read the CALLER's image, so there is no ordering between the two.
        const uint32_t dpr = L0.pitch >> 2;
        const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
        if (idx >= dpr * L0.h) return;
        const uint32_t y = idx / dpr, x0 = (idx - y * dpr) << 2;
        const uint8_t* __restrict__ row = src0 + (size_t)y * a.src_pitch;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (x0 + k < L0.w) packed |= (uint32_t)row[x0 + k] << (8 * k);
        *reinterpret_cast<uint32_t*>(arena + L0.offset + (size_t)y * L0.pitch + x0) = packed;
        return;
    }
    __shared__ float unorm[256];
    unorm[threadIdx.x] = (float)threadIdx.x / 255.0f;
    __syncthreads();
    const uint32_t bid = blockIdx.x - a.copy_blocks;
    // which level does this workgroup belong to?  (wave-uniform scan over <= 7 entries)
    int lv = 1;
#pragma unroll
    for (int i = 2; i < CLC_MAX_LEVELS; ++i)
        if (i < a.pd.levels && bid >= a.pd.blk_begin[i]) lv = i;
    const LevelDesc L = a.pd.lv[lv];
    const float gs = a.pd.f[lv];
    const uint32_t dpr = L.pitch >> 2;                       // dwords per output row
    const uint32_t idx = (bid - a.pd.blk_begin[lv]) * 256u + threadIdx.x;
    if (idx >= dpr * L.h) return;
    const uint32_t y = idx / dpr;
    const uint32_t x0 = (idx - y * dpr) << 2;
    const uint8_t* __restrict__ src = src0;
    const int W = (int)L0.w, H = (int)L0.h;

    const float fy = ((float)y + 0.5f) * gs - 0.5f;          // CUDALERP.cu:160
    const float fl_y = floorf(fy);
    const float wt_y = fy - fl_y;
    const float invwt_y = 1.0f - wt_y;
    const int j = (int)fl_y;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = x0 + k;
        const float fx = ((float)x + 0.5f) * gs - 0.5f;      // :165
        const float fl_x = floorf(fx);
        const int i = (int)fl_x;
        const float f_w = tap(src, a.src_pitch, W, H, i, j, unorm);
        const float f_z = tap(src, a.src_pitch, W, H, i + 1, j, unorm);
        const float f_x = tap(src, a.src_pitch, W, H, i, j + 1, unorm);
        const float f_y = tap(src, a.src_pitch, W, H, i + 1, j + 1, unorm);
        const float wt_x = fx - fl_x;
        const float invwt_x = 1.0f - wt_x;
        const float xa = invwt_x * f_w + wt_x * f_z;         // :172
        const float xb = invwt_x * f_x + wt_x * f_y;         // :173
        const float res = 255.0f * (invwt_y * xa + wt_y * xb) + 0.5f;   // :174
        const uint32_t v = x < L.w ? (uint32_t)(uint8_t)res : 0u;       // truncating store :177
        packed |= v << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(arena + L.offset + (size_t)y * L.pitch + x0) = packed;
}

hipError_t launch_pyramid_batch(const PyramidDesc& pd, uint8_t* arena, size_t slot_stride, const uint8_t* const* d_src,
                                int n_img, uint32_t src_pitch, hipStream_t stream, Profiler* prof)
{
    if (n_img <= 0) return hipSuccess;
    if (n_img > kMaxBatch || slot_stride > 0xFFFFFFFFull) return hipErrorInvalidValue;
    LerpArgs a;
    a.pd = pd;
    // d_src == level 0 inside the arena (single image only): nothing to copy
    const bool in_place = n_img == 1 && d_src[0] == arena + pd.lv[0].offset;
    a.src_pitch = in_place ? pd.lv[0].pitch : src_pitch;
    a.copy_blocks = in_place ? 0u : (pd.lv[0].pitch / 4 * pd.lv[0].h + 255) / 256;
    a.slot_stride = (uint32_t)slot_stride;
    for (int b = 0; b < kMaxBatch; ++b) a.src[b] = b < n_img ? d_src[b] : nullptr;
    const uint32_t nblk = a.copy_blocks + (pd.levels > 1 ? pd.blk_begin[pd.levels] : 0u);
    if (nblk == 0) return hipSuccess;
    prof_mark(prof, CLC_KERNEL_PYRAMID, true, stream);
    hipLaunchKernelGGL(pyramid_kernel, dim3(nblk, (uint32_t)n_img), dim3(256), 0, stream, a, arena);
    prof_mark(prof, CLC_KERNEL_PYRAMID, false, stream);
    return hipGetLastError();
}

hipError_t launch_pyramid(const PyramidDesc& pd, uint8_t* arena, const uint8_t* d_src, uint32_t src_pitch,
                          hipStream_t stream, Profiler* prof)
{
    return launch_pyramid_batch(pd, arena, 0, &d_src, 1, src_pitch, stream, prof);
}

} // namespace clc
