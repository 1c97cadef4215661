// valu_rate.hip -- issue-rate microbenchmark for the VALU ops the hot path is made of (not shipped).
// Each wave runs ITER x 32 independent instructions of one kind; 8 waves per SIMD, every CU busy.
// Reports wave-instructions per SIMD per cycle using the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, int iters, unsigned long long* clk)
{
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x;
    uint32_t s = blockIdx.x * 7u + 1u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define OPS(I) \
        if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[I]) : "s"(s)); \
        else if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 2) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 3) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 4) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 5) asm volatile("v_mad_u32_u24 %0, %1, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 6) asm volatile("v_med3_u32 %0, %1, %0, %2" : "+v"(r[I]) : "v"(r[(I + 1) & 15]), "v"(r[(I + 2) & 15])); \
        else if (OP == 7) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[I])); \
        else if (OP == 8) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 9) asm volatile("v_min_u32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 10) asm volatile("v_lshl_add_u32 %0, %1, 22, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 11) asm volatile("v_alignbyte_b32 %0, %1, %0, 1" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 12) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 13) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 14) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 15) asm volatile("v_or_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 16) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 17) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[I])); \
        else if (OP == 18) asm volatile("v_max_u32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 19) asm volatile("v_add3_u32 %0, %1, %0, %2" : "+v"(r[I]) : "v"(r[(I + 1) & 15]), "v"(r[(I + 2) & 15])); \
        else if (OP == 20) asm volatile("v_xad_u32 %0, %1, %0, %2" : "+v"(r[I]) : "v"(r[(I + 1) & 15]), "v"(r[(I + 2) & 15])); \
        else if (OP == 21) asm volatile("v_mov_b32 %0, %1" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 22) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[I]) : "s"(s)); \
        else if (OP == 23) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 24) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 25) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[I])); \
        else if (OP == 26) asm volatile("v_floor_f32 %0, %0" : "+v"(r[I])); \
        else if (OP == 27) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(r[I])); \
        else if (OP == 28) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 29) asm volatile("v_xnor_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 30) asm volatile("v_bitop3_b32 %0, %1, %0, %2 bitop3:0x96" : "+v"(r[I]) : "v"(r[(I + 1) & 15]), "v"(r[(I + 2) & 15])); \
        else if (OP == 31) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15]), "v"(r[(I + 2) & 15])); \
        else if (OP == 32) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 33) asm volatile("v_min_i32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 34) asm volatile("v_subrev_u32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 35) asm volatile("v_mad_i32_i24 %0, %1, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 36) { if ((I) & 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); } \
        else if (OP == 37) { if ((I) & 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 15) & 15])); else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 2) & 15])); } \
        else if (OP == 38) { if ((I) & 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[I]) : "s"(s)); } \
        else if (OP == 40) asm volatile("v_xor_b32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 41) { if ((I) & 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 15) & 15])); else asm volatile("v_xor_b32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(r[I]) : "v"(r[(I + 2) & 15]), "v"(r[(I + 4) & 15])); } \
        else if (OP == 42) asm volatile("v_mov_b32 %0, %1" : "+v"(r[I]) : "s"(s)); \
        else if (OP == 43) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); \
        else if (OP == 44) { uint32_t tmp; asm volatile("v_xor_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(r[14 + ((I) & 1)]), "=&v"(tmp) : "v"(r[(I) % 7]), "v"(r[7 + (I) % 7])); } \
        else if (OP == 45) { uint32_t tmp; asm volatile("v_xor_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(r[12 + ((I) & 3)]), "=&v"(tmp) : "v"(r[(I) % 6]), "v"(r[6 + (I) % 6])); } \
        else if (OP == 46) { uint32_t tmp; asm volatile("v_xor_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(r[14 + ((I) & 1)]), "=&v"(tmp) : "s"(s), "v"(r[(I) % 7])); } \
        else if (OP == 47) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r[15]) : "v"(r[(I) % 7]), "v"(r[7 + (I) % 7])); \
        else if (OP == 48) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(r[12 + ((I) & 3)]) : "v"(r[(I) % 6]), "v"(r[6 + (I) % 6])); \
        else if (OP == 49) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(r[12 + ((I) & 3)]) : "v"(r[(I) % 12])); \
        else if (OP == 50) asm volatile("v_add_f32 %0, %1, %0\n\tv_cvt_i32_f32 %0, %0\n\tv_med3_i32 %0, %0, 0, %2" : "+v"(r[(I)]) : "v"(r[(I + 1) & 15]), "v"(r[(I + 2) & 15])); \
        else if (OP == 39) { if (((I) & 3) == 3) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); else asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15])); }
        OPS(0) OPS(1) OPS(2) OPS(3) OPS(4) OPS(5) OPS(6) OPS(7) OPS(8) OPS(9) OPS(10) OPS(11) OPS(12) OPS(13) OPS(14) OPS(15)
        OPS(0) OPS(1) OPS(2) OPS(3) OPS(4) OPS(5) OPS(6) OPS(7) OPS(8) OPS(9) OPS(10) OPS(11) OPS(12) OPS(13) OPS(14) OPS(15)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = rt1 - rt0; }
}

// packed fp32 FMA: 2 flops-pairs per lane per instruction
__global__ __launch_bounds__(256) void rate_pk(float2* out, int iters, unsigned long long* clk)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = f2{ (float)threadIdx.x + i, 1.0f + i };
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#define PK(I) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(r[I]) : "v"(r[(I + 1) & 15]));
        PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) PK(8) PK(9) PK(10) PK(11) PK(12) PK(13) PK(14) PK(15)
        PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) PK(8) PK(9) PK(10) PK(11) PK(12) PK(13) PK(14) PK(15)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    f2 acc = r[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) acc += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = make_float2(acc.x, acc.y);
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = rt1 - rt0; }
}

template <typename F>
void bench(const char* name, F launch, int blocks, int iters, unsigned long long* dclk)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(2 * blocks);
    CHECK(hipMemcpy(h.data(), dclk, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; rt += h[2 * b + 1]; }
    cyc /= blocks; rt /= blocks;
    const double ghz = cyc / (rt * 10.0);               // realtime counter = 100 MHz
    const double winstr_per_simd = (double)blocks * 4 /*waves per block*/ * iters * 32 / 1024.0;
    const double wall_cycles = ms * 1e-3 * ghz * 1e9;
    printf("%-18s wall %8.1f us  clock %.2f GHz  cycles/wave-instr/SIMD %.2f  lane-ops/clk/CU %.1f\n",
           name, ms * 1e3, ghz, wall_cycles / winstr_per_simd, winstr_per_simd * 64 * 4 / wall_cycles);
}

int main()
{
    const int blocks = 2048, iters = 2000;   // 2048 blocks x 4 waves = 8 waves per SIMD, one round
    uint32_t* dout; unsigned long long* dclk;
    CHECK(hipMalloc((void**)&dout, (size_t)blocks * 256 * 8)); CHECK(hipMalloc((void**)&dclk, blocks * 16));
    const char* names[] = { "v_xor_b32(s,v)", "v_bcnt_u32_b32", "v_add_f32", "v_fma_f32", "v_dot4_u32_u8", "v_mad_u32_u24", "v_med3_u32",
                            "v_cvt_i32_f32", "v_add_u32", "v_min_u32", "v_lshl_add_u32", "v_alignbyte_b32", "v_mul_f32",
                            "v_xor_b32(v,v)", "v_and_b32", "v_or_b32", "v_sub_u32", "v_lshlrev_b32", "v_max_u32", "v_add3_u32", "v_xad_u32", "v_mov_b32",
                            "v_add_u32(s,v)", "v_max_f32", "v_sub_f32", "v_cvt_f32_i32", "v_floor_f32", "v_bfe_u32", "v_cndmask_b32", "v_xnor_b32",
                            "v_bitop3_b32", "v_sad_u8", "v_mul_u32_u24", "v_min_i32", "v_subrev_u32", "v_mad_i32_i24",
                            "alt xor(vv)/bcnt indep", "alt xor(vv)->bcnt dep", "alt xor(sv)/bcnt", "3 xor(vv) : 1 bcnt",
                            "v_xor_b32_dpp newbcast", "alt xor_dpp->bcnt dep", "v_mov_b32(s)", "v_mov_b32_dpp newbcast",
                            "k2nn pairs vv, 2 chains", "k2nn pairs vv, 4 chains", "k2nn pairs sv, 2 chains",
                            "dot4 one acc chain", "dot4 four acc chains", "dot4 a*a four chains", "add_f32->cvt->med3 run (x3)" };
#define B(OP) bench(names[OP], [&]() { hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, dout, iters, dclk); }, blocks, iters, dclk);
    B(0) B(1) B(2) B(3) B(4) B(5) B(6) B(7) B(8) B(9) B(10) B(11) B(12) B(13) B(14) B(15) B(16) B(17) B(18) B(19) B(20) B(21) B(22) B(23) B(24) B(25) B(26) B(27) B(28) B(29) B(30) B(31) B(32) B(33) B(34) B(35) B(36) B(37) B(38) B(39) B(40) B(41) B(42) B(43) B(44) B(45) B(46) B(47) B(48) B(49) B(50)
    bench("v_pk_fma_f32", [&]() { hipLaunchKernelGGL(rate_pk, dim3(blocks), dim3(256), 0, 0, (float2*)dout, iters, dclk); }, blocks, iters, dclk);
    return 0;
}
