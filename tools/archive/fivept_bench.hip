// tools/fivept_bench.hip -- stand-alone timing + hit-rate harness for the five-point solver of csrc/fivept.h
// (phase stamps via FPT_STAMP).  Build: hipcc -O3 --offload-arch=gfx950 -o /tmp/fivept_bench tools/fivept_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
#include "fivept_scene.h"

__device__ unsigned long long* g_stamps;
#define FPT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = __builtin_readcyclecounter(); } while (0)
#include "../coloc_amd/csrc/fivept_wave.h"

__global__ __launch_bounds__(64) void bench_kernel(const double* __restrict__ q1g, const double* __restrict__ q2g, const int S,
                                                   double* __restrict__ Eout, int* __restrict__ nout, unsigned long long* stamps)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps = stamps;
    __syncthreads();
    const int sidx = blockIdx.x;
    double q1[5][2], q2[5][2];
    for (int p = 0; p < 5; ++p) for (int c = 0; c < 2; ++c) { q1[p][c] = q1g[10 * sidx + 2 * p + c]; q2[p][c] = q2g[10 * sidx + 2 * p + c]; }
    __shared__ FptWorkspace ws;
    __shared__ double E[90];
    FPT_STAMP(0);
    const int n = fpw::solve(q1, q2, E, ws);
    FPT_STAMP(9);
    __syncthreads();
    if (threadIdx.x == 0) nout[sidx] = n;
    for (int i = threadIdx.x; i < 90; i += 64) Eout[90 * sidx + i] = i < 9 * n ? E[i] : 0.0;
}

int main(int argc, char** argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 256;
    std::vector<double> q1, q2, Etrue;
    fpt_make_scenes(S, q1, q2, Etrue);
    double *d1, *d2, *dE; int* dn; unsigned long long* dst;
    hipMalloc(&d1, sizeof(double) * 10 * S); hipMalloc(&d2, sizeof(double) * 10 * S); hipMalloc(&dE, sizeof(double) * 90 * S);
    hipMalloc(&dn, sizeof(int) * S); hipMalloc(&dst, 16 * 8);
    hipMemcpy(d1, q1.data(), sizeof(double) * 10 * S, hipMemcpyHostToDevice);
    hipMemcpy(d2, q2.data(), sizeof(double) * 10 * S, hipMemcpyHostToDevice);
    hipMemset(dst, 0, 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(bench_kernel, dim3(S), dim3(64), 0, 0, d1, d2, S, dE, dn, dst);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(bench_kernel, dim3(S), dim3(64), 0, 0, d1, d2, S, dE, dn, dst);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> E(90 * S); std::vector<int> n(S); unsigned long long st[16];
    hipMemcpy(E.data(), dE, sizeof(double) * 90 * S, hipMemcpyDeviceToHost);
    hipMemcpy(n.data(), dn, sizeof(int) * S, hipMemcpyDeviceToHost);
    hipMemcpy(st, dst, 128, hipMemcpyDeviceToHost);
    int hits = 0; long nsol = 0; uint64_t h = 0;
    fpt_score(S, E, n, Etrue, &hits, &nsol, &h);
    printf("S=%d  %.1f us per launch  hit-rate %.4f  solutions/sample %.2f  checksum %016llx\n", S, 1e3 * ms / reps, (double)hits / S, (double)nsol / S,
           (unsigned long long)h);
    printf("block 0 phases (cycles): ");
    for (int i = 1; i < 16; ++i) printf("[%d] %lld  ", i, (long long)(st[i] - st[0]));
    printf("\n");
    return 0;
}
