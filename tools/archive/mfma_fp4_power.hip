// mfma_fp4_power.hip -- does the VALUE pattern of the FP4 operands change the time of a sustained v_mfma_scale_f32_32x32x64_f8f6f4 loop?
// (round 4: the K2NN sweep holds 1.83 GHz with its +-1 x +-1 operands, 2.16 GHz without the MFMAs: is part of that the multipliers' switching?)
// 3 waves per SIMD, two chains of 8 MFMAs per "tile", operands cycling through 8 register sets per side (like the sweep's k-steps);
// every pattern runs back to back for ~1 s so that the power management has settled, then 200 launches are timed with events.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_fp4_power tools/mfma_fp4_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <chrono>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// mode: 0 = +-1 (0x2 | bit << 3), 1 = {0, 1} (bit << 1), 2 = all +1, 3 = all 0, 4 = {0,1} with 25 % ones
__device__ __forceinline__ uint32_t make_operand(uint32_t r, int mode)
{
    switch (mode) {
    case 0: return (r & 0x88888888u) | 0x22222222u;
    case 1: return r & 0x22222222u;
    case 2: return 0x22222222u;
    case 3: return 0u;
    default: return r & (r >> 1) & 0x22222222u;
    }
}

__global__ __launch_bounds__(256) void power_kernel(float* out, const int tiles, const int mode_a, const int mode_b, const uint32_t seed)
{
    v8i a[8], b[2][8];
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = v8i{ 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < 4; ++k) a[j][k] = (int)make_operand(hash32(seed + t * 64u + j * 4u + k), mode_a);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            b[c][j] = v8i{ 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 4; ++k) b[c][j][k] = (int)make_operand(hash32(seed * 3u + t * 64u + 32u + c * 32u + j * 4u + k), mode_b);
        }
    }
    v16f acc[2];
    float best = 3.0e38f;
    for (int it = 0; it < tiles; ++it) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][i] = 8388608.0f + (float)it;   // (nothing is loop invariant)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[j], b[c][j], acc[c], 4, 4, 0, 0x8B8B8B8B, 0, 0x7F7F7F7F);
#pragma unroll
            for (int i = 0; i < 16; ++i) best = fminf(best, acc[c][i]);
        }
    }
    if (best == 12345.678f) out[t] = best;
}

int main()
{
    float* d; CHECK(hipMalloc(&d, 4 * 256 * 768));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int tiles = 17 * 8;                      // ~ 8 x the sweep's loop: 2 176 MFMAs per wave
    struct { const char* name; int ma, mb; } pat[] = {
        { "A +-1      x B +-1      (the sweep today)", 0, 0 },
        { "A {0,1}    x B +-1", 1, 0 },
        { "A {0,1}    x B {0,1}", 1, 1 },
        { "A {0,1}25% x B {0,1}25%", 4, 4 },
        { "A all +1   x B all +1   (no toggling)", 2, 2 },
        { "A all 0    x B all 0", 3, 3 },
        { "A +-1      x B +-1      (again)", 0, 0 },
    };
    for (auto& p : pat) {
        const auto t0 = std::chrono::steady_clock::now();
        int n = 0;
        while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(1500)) {
            for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(power_kernel, dim3(768), dim3(256), 0, 0, d, tiles, p.ma, p.mb, 17u + n);
            CHECK(hipDeviceSynchronize());
            n += 50;
        }
        CHECK(hipEventRecord(e0));
        for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(power_kernel, dim3(768), dim3(256), 0, 0, d, tiles, p.ma, p.mb, 1000u + k);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / 200.0;
        const double mfma_per_simd = 3.0 * tiles * 16;             // 3 waves per SIMD
        const double ghz_equiv = mfma_per_simd * 32.0 / (us * 1e3);   // clock at which a 100 % busy pipe would take this long
        printf("%-44s %8.2f us per launch   %.3f GHz-equivalent (32 cycles per MFMA, pipe 100 %% busy)\n", p.name, us, ghz_equiv);
        fflush(stdout);
    }
    return 0;
}
