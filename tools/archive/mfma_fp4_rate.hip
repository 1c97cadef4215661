// mfma_fp4_rate.hip -- issue rate of the two FP4 block-scaled MFMA shapes (operands in registers, N independent accumulator chains per wave),
// 1..3 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_fp4_rate tools/mfma_fp4_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SHAPE, int CHAINS>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t* out, const int iters, const int seed)
{
    v8i a = { seed, seed + 1, seed + 2, seed + 3, 0, 0, 0, 0 }, b = { seed * 3, seed * 5, seed * 7, seed * 11, 0, 0, 0, 0 };
    a[0] += threadIdx.x; b[1] += threadIdx.x;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    if (SHAPE == 32) {
        v16f acc[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) sink += acc[c][i];
    } else {
        v4f acc[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[c], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 4; ++i) sink += acc[c][i];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (sink == 12345.678f) out[blockIdx.x] = 0;
}

template <int SHAPE, int CHAINS>
static int run(const char* name, int waves_per_simd)
{
    uint64_t* d; CHECK(hipMalloc(&d, 8 * 4096));
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;      // 256-thread blocks = 4 waves = one per SIMD; waves_per_simd blocks per CU
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((rate_kernel<SHAPE, CHAINS>), dim3(blocks), dim3(256), 0, 0, d, iters, rep + 1);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<uint64_t> h(blocks); CHECK(hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += (double)v; s /= blocks;
    const double per = s / ((double)iters * CHAINS);                      // cycles per MFMA per wave
    const double flop = SHAPE == 32 ? 2.0 * 32 * 32 * 64 : 2.0 * 16 * 16 * 128;
    printf("  %-34s %d wave(s)/SIMD: %7.2f cycles per MFMA and wave -> %6.1f cycles per MFMA on the SIMD, %7.1f flop/cycle/SIMD\n", name, waves_per_simd, per,
           per / waves_per_simd, flop / (per / waves_per_simd));
    CHECK(hipFree(d));
    return 0;
}

int main()
{
    for (int w = 1; w <= 3; ++w) {
        if (run<32, 1>("32x32x64 fp4, 1 dependent chain", w)) return 1;
        if (run<32, 2>("32x32x64 fp4, 2 chains", w)) return 1;
        if (run<32, 4>("32x32x64 fp4, 4 chains", w)) return 1;
        if (run<16, 1>("16x16x128 fp4, 1 dependent chain", w)) return 1;
        if (run<16, 2>("16x16x128 fp4, 2 chains", w)) return 1;
        if (run<16, 4>("16x16x128 fp4, 4 chains", w)) return 1;
        if (run<16, 8>("16x16x128 fp4, 8 chains", w)) return 1;
    }
    return 0;
}
