// dispatch_rate.hip -- how fast can the chip start and retire one-wave workgroups of a given resource footprint?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/dispatch_rate tools/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LDS_BYTES, int VGPRS, int SPIN>
__global__ __launch_bounds__(64) void k(uint32_t* out)
{
    __shared__ uint8_t lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
    uint32_t v[VGPRS];
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) v[i] = threadIdx.x * (i + 1);
    if (LDS_BYTES > 0) { lds[threadIdx.x * 4] = (uint8_t)blockIdx.x; lds[LDS_BYTES - 1 - threadIdx.x] = 1; }
    // SPIN dependent VALU ops per register: keeps the registers live and sets the wave's lifetime
    for (int s = 0; s < SPIN; ++s) {
#pragma unroll
        for (int i = 0; i < VGPRS; ++i) v[i] = v[i] * 1664525u + 1013904223u;
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < VGPRS; ++i) acc ^= v[i];
    if (LDS_BYTES > 0) acc += lds[(threadIdx.x * 4 + 4) % LDS_BYTES];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

template <typename F> static void run(const char* name, F launch, int n)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < 15; ++i) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    CHECK(hipEventRecord(e0)); for (int i = 0; i < 50; ++i) launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s n=%5d single %7.2f us  sustained %7.2f us  -> %.2f ns per workgroup\n", name, n, ts[7] * 1e3, ms / 50 * 1e3, ms / 50 * 1e6 / n);
}

int main()
{
    uint32_t* d; CHECK(hipMalloc((void**)&d, 65536 * 4));
    for (int n : { 10000, 20000 }) {
#define RUN(L, V, S) run("lds " #L " B, " #V " live vgprs, spin " #S, [&]() { hipLaunchKernelGGL((k<L, V, S>), dim3(n), dim3(64), 0, 0, d); }, n)
        RUN(0, 8, 0); RUN(12688, 8, 0); RUN(0, 140, 0); RUN(12688, 140, 0); RUN(6000, 140, 0); RUN(12688, 100, 0);
        RUN(12688, 140, 4); RUN(12688, 140, 16); RUN(0, 140, 16); RUN(12688, 8, 256);
    }
    return 0;
}
