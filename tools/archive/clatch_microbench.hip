// clatch_microbench.hip -- times clatch_kernel variants (compile-time -D switches) on synthetic data.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-DCLATCH_...] -o tools/clmb tools/clatch_microbench.hip
#include "../coloc_amd/csrc/clatch.hip"
#ifndef VARIANT
#define VARIANT "production"
#endif
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
namespace clc { void prof_mark(Profiler*, int, bool, hipStream_t) {} }
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main()
{
    using namespace clc;
    const uint32_t W = 640, H = 480; const int n = 10000;
    PyramidDesc pd{}; pd.levels = 8; float f = 1.f; uint32_t off = 0;
    for (int i = 0; i < 8; ++i) {
        if (i) f *= 1.2f;
        pd.lv[i].w = i ? (uint32_t)((float)W / f + 0.5f) : W; pd.lv[i].h = i ? (uint32_t)((float)H / f + 0.5f) : H;
        pd.lv[i].pitch = (pd.lv[i].w + 63) / 64 * 64; pd.lv[i].offset = off; off += (pd.lv[i].pitch * pd.lv[i].h + 255) / 256 * 256; pd.f[i] = f;
    }
    std::vector<uint8_t> harena(off + 256);
    uint64_t s = 88172645463325252ull; auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (auto& v : harena) v = (uint8_t)rnd();
    std::vector<clc_keypoint> hk(n);
    for (auto& k : hk) { k.scale = rnd() % 8; k.x = 3 + rnd() % (pd.lv[k.scale].w - 6); k.y = 3 + rnd() % (pd.lv[k.scale].h - 6); k.angle = ((rnd() % 62832) - 31416) * 1e-4f; k.score = 0; }
    uint8_t* darena; clc_keypoint* dk; uint64_t* dd;
    CHECK(hipMalloc((void**)&darena, harena.size())); CHECK(hipMalloc((void**)&dk, n * sizeof(clc_keypoint))); CHECK(hipMalloc((void**)&dd, (size_t)n * 64));
    CHECK(hipMemcpy(darena, harena.data(), harena.size(), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dk, hk.data(), n * sizeof(clc_keypoint), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CHECK(launch_clatch(pd, darena, dk, n, dd, 0, nullptr));
    CHECK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < 20; ++i) { CHECK(hipEventRecord(e0)); CHECK(launch_clatch(pd, darena, dk, n, dd, 0, nullptr)); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    std::vector<uint64_t> hd((size_t)n * 8); CHECK(hipMemcpy(hd.data(), dd, hd.size() * 8, hipMemcpyDeviceToHost));
    uint64_t chk = 0; for (auto v : hd) chk = chk * 1315423911ull + v;
    printf("%-40s median %8.1f us  min %8.1f us  (%.1f Mdesc/s)  checksum %016llx\n", VARIANT, ts[10] * 1e3, ts[0] * 1e3, n / (ts[10] * 1e3), (unsigned long long)chk);
    return 0;
}
