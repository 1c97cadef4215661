// k2nn_microbench.hip -- standalone experiment harness for the K2NN sweep inner loop (not shipped).
// Variants share the key/top-2 logic of coloc_amd/csrc/k2nn.hip; they differ in how the
// wave-uniform train vector reaches the VALU.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/k2mb tools/k2nn_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const u32x4 __attribute__((address_space(4)))* const_u4_ptr;
typedef const u32x4 __attribute__((address_space(1)))* global_cu4_ptr;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint32_t bcnt_first(uint32_t x) { uint32_t r; asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) { uint32_t r; asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc)); return r; }
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int R>
__device__ __forceinline__ void sweep_one(const uint32_t (&q)[R][16], const u32x4 a, const u32x4 b, const u32x4 c, const u32x4 d,
                                          const uint32_t t_rel, uint32_t (&best)[R], uint32_t (&second)[R])
{
    const uint32_t tw[16] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t acc = bcnt_first(q[r][0] ^ tw[0]);
#pragma unroll
        for (int k = 1; k < 16; ++k) acc = bcnt_acc(q[r][k] ^ tw[k], acc);
        const uint32_t key = (acc << 22) + t_rel;
        second[r] = umed3(best[r], second[r], key);
        best[r] = min(best[r], key);
    }
}

template <int R>
__device__ __forceinline__ void sweep_one_paired(const uint32_t (&q)[R][16], const u32x4 a, const u32x4 b, const u32x4 c, const u32x4 d,
                                                 const uint32_t t_rel, uint32_t (&best)[R], uint32_t (&second)[R])
{
    const uint32_t tw[16] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t acc = 0, tmp;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            asm volatile("v_xor_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(acc), "=&v"(tmp) : "v"(q[r][k]), "v"(tw[k]));
        const uint32_t key = (acc << 22) + t_rel;
        second[r] = umed3(best[r], second[r], key);
        best[r] = min(best[r], key);
    }
}

// paired asm, the R chains interleaved word by word (ILP inside one wave), optional second accumulator
template <int R, int NACC>
__device__ __forceinline__ void sweep_one_paired_il(const uint32_t (&q)[R][16], const u32x4 a, const u32x4 b, const u32x4 c, const u32x4 d,
                                                    const uint32_t t_rel, uint32_t (&best)[R], uint32_t (&second)[R])
{
    const uint32_t tw[16] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
    uint32_t acc[R][NACC], tmp;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[r][n] = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int r = 0; r < R; ++r)
            asm volatile("v_xor_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[r][k % NACC]), "=&v"(tmp) : "v"(q[r][k]), "v"(tw[k]));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t d0 = acc[r][0];
#pragma unroll
        for (int n = 1; n < NACC; ++n) d0 += acc[r][n];
        const uint32_t key = (d0 << 22) + t_rel;
        second[r] = umed3(best[r], second[r], key);
        best[r] = min(best[r], key);
    }
}

// paired asm with the train word as SGPR operand: v_xor_b32 v, s, v directly followed by its v_bcnt
template <int R, int IL>
__device__ __forceinline__ void sweep_one_paired_sgpr(const uint32_t (&q)[R][16], const u32x4 a, const u32x4 b, const u32x4 c, const u32x4 d,
                                                      const uint32_t t_rel, uint32_t (&best)[R], uint32_t (&second)[R])
{
    const uint32_t tw[16] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
    uint32_t acc[R], tmp;
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0;
    if (IL) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r)
                asm volatile("v_xor_b32 %1, %3, %2\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[r]), "=&v"(tmp) : "v"(q[r][k]), "s"(tw[k]));
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("v_xor_b32 %1, %3, %2\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[r]), "=&v"(tmp) : "v"(q[r][k]), "s"(tw[k]));
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t key = (acc[r] << 22) + t_rel;
        second[r] = umed3(best[r], second[r], key);
        best[r] = min(best[r], key);
    }
}

// MODE 0: plain loop (s_load, wait, compute).  MODE 1: no loads in the loop (VALU ceiling).
// MODE 2: explicit double buffer: wait, issue next s_load, compute current.
template <int R, int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES) void sweep(const u32x4* __restrict__ Q, int nq, const u32x4* __restrict__ T, int nt,
                                                   int splits, int t_per_split, u32x2* __restrict__ partial, int nq_pad)
{
    const uint32_t qblock = blockIdx.x / splits;
    const uint32_t split = blockIdx.x - qblock * splits;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t qbase = qblock * (64u * R * WAVES) + wave * (64u * R) + lane;
    uint32_t q[R][16];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t qi = qbase + 64u * r; if (qi >= (uint32_t)nq) qi = nq - 1;
        const global_cu4_ptr qp = (global_cu4_ptr)(uintptr_t)Q + (size_t)qi * 4u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const u32x4 v = qp[k]; q[r][4*k] = v.x; q[r][4*k+1] = v.y; q[r][4*k+2] = v.z; q[r][4*k+3] = v.w; }
    }
    uint32_t best[R], second[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = 0xFFFFFFFFu; second[r] = 0xFFFFFFFFu; }
    const uint32_t t0 = split * t_per_split;
    uint32_t t1 = t0 + t_per_split; if (t1 > (uint32_t)nt) t1 = nt;
    const_u4_ptr tp = (const_u4_ptr)(uintptr_t)T + (size_t)t0 * 4u;
    if (MODE == 0) {
        for (uint32_t t = t0; t < t1; ++t, tp += 4) {
            const u32x4 a = tp[0], b = tp[1], c = tp[2], d = tp[3];
            sweep_one<R>(q, a, b, c, d, t - t0, best, second);
        }
    } else if (MODE == 1) {
        const u32x4 a = tp[0], b = tp[1], c = tp[2], d = tp[3];
        for (uint32_t t = t0; t < t1; ++t) {
            sweep_one<R>(q, a, b, c, d, t - t0, best, second);
            asm volatile("" ::: "memory");
        }
    } else {
        u32x4 a = tp[0], b = tp[1], c = tp[2], d = tp[3];
        for (uint32_t t = t0; t < t1; ++t) {
            tp += 4;
            __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0): current vector has landed
            __builtin_amdgcn_sched_barrier(0);
            const_u4_ptr np = (t + 1 < t1) ? tp : tp - 4;   // never read past the split
            const u32x4 na = np[0], nb = np[1], nc = np[2], nd = np[3];
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 9) sweep_one_paired_sgpr<R, 0>(q, a, b, c, d, t - t0, best, second);
            else if (MODE == 10) sweep_one_paired_sgpr<R, 1>(q, a, b, c, d, t - t0, best, second);
            else sweep_one<R>(q, a, b, c, d, t - t0, best, second);
            __builtin_amdgcn_sched_barrier(0);
            a = na; b = nb; c = nc; d = nd;
        }
    }
    u32x2 __attribute__((address_space(1)))* prow = (u32x2 __attribute__((address_space(1)))*)(uintptr_t)partial + (size_t)split * nq_pad;
#pragma unroll
    for (int r = 0; r < R; ++r) { const uint32_t qi = qbase + 64u * r; if (qi < (uint32_t)nq) prow[qi] = u32x2{ best[r], second[r] }; }
}

// MODE 3: train tiles staged in LDS (double buffered), broadcast-read into VGPRs -> v_xor_b32 v,v,v (2-cycle form)
template <int R, int WAVES, int TT, bool PAIRED = false>
__global__ __launch_bounds__(64 * WAVES) void sweep_lds(const u32x4* __restrict__ Q, int nq, const u32x4* __restrict__ T, int nt,
                                                       int splits, int t_per_split, u32x2* __restrict__ partial, int nq_pad)
{
    __shared__ u32x4 tile[2][TT * 4];
    const uint32_t qblock = blockIdx.x / splits;
    const uint32_t split = blockIdx.x - qblock * splits;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t qbase = qblock * (64u * R * WAVES) + wave * (64u * R) + lane;
    uint32_t q[R][16];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t qi = qbase + 64u * r; if (qi >= (uint32_t)nq) qi = nq - 1;
        const global_cu4_ptr qp = (global_cu4_ptr)(uintptr_t)Q + (size_t)qi * 4u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const u32x4 v = qp[k]; q[r][4*k] = v.x; q[r][4*k+1] = v.y; q[r][4*k+2] = v.z; q[r][4*k+3] = v.w; }
    }
    uint32_t best[R], second[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = 0xFFFFFFFFu; second[r] = 0xFFFFFFFFu; }
    const uint32_t t0 = split * t_per_split;
    uint32_t t1 = t0 + t_per_split; if (t1 > (uint32_t)nt) t1 = nt;
    const global_cu4_ptr tg = (global_cu4_ptr)(uintptr_t)T;
    constexpr int NT = 64 * WAVES;                 // threads
    constexpr int PER = (TT * 4 + NT - 1) / NT;    // u32x4 per thread per tile
    const uint32_t ntiles = (t1 - t0 + TT - 1) / TT;
    u32x4 stage[PER];
    auto gload = [&](uint32_t tile_i) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const uint32_t e = threadIdx.x + p * NT;             // u32x4 index inside the tile
            uint32_t row = t0 + tile_i * TT + (e >> 2);
            if (row >= t1) row = t1 - 1;
            stage[p] = (e < TT * 4) ? tg[(size_t)row * 4u + (e & 3u)] : u32x4{0, 0, 0, 0};
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PER; ++p) { const uint32_t e = threadIdx.x + p * NT; if (e < TT * 4) tile[buf][e] = stage[p]; }
    };
    gload(0); lstore(0);
    __syncthreads();
    for (uint32_t ti = 0; ti < ntiles; ++ti) {
        const int buf = ti & 1;
        if (ti + 1 < ntiles) gload(ti + 1);
        const uint32_t base = t0 + ti * TT;
        const uint32_t cnt = min((uint32_t)TT, t1 - base);
        for (uint32_t j = 0; j < cnt; ++j) {
            const u32x4 a = tile[buf][j * 4], b = tile[buf][j * 4 + 1], c = tile[buf][j * 4 + 2], d = tile[buf][j * 4 + 3];
            if (PAIRED) sweep_one_paired<R>(q, a, b, c, d, base + j - t0, best, second);
            else sweep_one<R>(q, a, b, c, d, base + j - t0, best, second);
        }
        if (ti + 1 < ntiles) lstore(buf ^ 1);
        __syncthreads();
    }
    u32x2 __attribute__((address_space(1)))* prow = (u32x2 __attribute__((address_space(1)))*)(uintptr_t)partial + (size_t)split * nq_pad;
#pragma unroll
    for (int r = 0; r < R; ++r) { const uint32_t qi = qbase + 64u * r; if (qi < (uint32_t)nq) prow[qi] = u32x2{ best[r], second[r] }; }
}

static unsigned long long g_checksum = 0;
static constexpr int kMaxSplits = 1024;   // capacity of the partial buffer, in split rows
// MODE 6/7: WAVE-PRIVATE LDS staging (no workgroup barrier): each wave pulls 16 train vectors with one coalesced
// global_load_dwordx4, parks them in its own 2 x 1 KiB LDS ring and broadcast-reads them into VGPRs, so v_xor_b32
// runs in its 2-cycle VGPR form directly followed by the dependent v_bcnt (paired asm).
template <int R, int WAVES, int VAR = 0>
__global__ __launch_bounds__(64 * WAVES) void sweep_wlds(const u32x4* __restrict__ Q, int nq, const u32x4* __restrict__ T, int nt,
                                                        int splits, int t_per_split, u32x2* __restrict__ partial, int nq_pad)
{
    __shared__ u32x4 ring[WAVES][2][64];
    const uint32_t qblock = blockIdx.x / splits;
    const uint32_t split = blockIdx.x - qblock * splits;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t qbase = qblock * (64u * R * WAVES) + wave * (64u * R) + lane;
    uint32_t q[R][16];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t qi = qbase + 64u * r; if (qi >= (uint32_t)nq) qi = nq - 1;
        const global_cu4_ptr qp = (global_cu4_ptr)(uintptr_t)Q + (size_t)qi * 4u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const u32x4 v = qp[k]; q[r][4*k] = v.x; q[r][4*k+1] = v.y; q[r][4*k+2] = v.z; q[r][4*k+3] = v.w; }
    }
    uint32_t best[R], second[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = 0xFFFFFFFFu; second[r] = 0xFFFFFFFFu; }
    const uint32_t t0 = split * t_per_split;
    uint32_t t1 = t0 + t_per_split; if (t1 > (uint32_t)nt) t1 = nt;
    const global_cu4_ptr tg = (global_cu4_ptr)(uintptr_t)T;
    const uint32_t ntile = (t1 - t0 + 15u) / 16u;
    auto gload = [&](uint32_t ti) {
        uint32_t row = t0 + ti * 16u + (lane >> 2);
        if (row >= t1) row = t1 - 1u;
        return tg[(size_t)row * 4u + (lane & 3u)];
    };
    u32x4 stage = gload(0);
    for (uint32_t ti = 0; ti < ntile; ++ti) {
        u32x4* buf = ring[wave][ti & 1u];
        buf[lane] = stage;                                       // ds_write_b128, wave-private: no barrier needed
        if (ti + 1 < ntile) stage = gload(ti + 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t base = t0 + ti * 16u;
        const uint32_t cnt = min(16u, t1 - base);
        for (uint32_t j = 0; j < cnt; ++j) {
            const u32x4 a = buf[j * 4], b = buf[j * 4 + 1], c = buf[j * 4 + 2], d = buf[j * 4 + 3];
            if (VAR == 0) sweep_one_paired<R>(q, a, b, c, d, base + j - t0, best, second);
            else if (VAR == 1) sweep_one_paired_il<R, 1>(q, a, b, c, d, base + j - t0, best, second);
            else sweep_one_paired_il<R, 2>(q, a, b, c, d, base + j - t0, best, second);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    u32x2 __attribute__((address_space(1)))* prow = (u32x2 __attribute__((address_space(1)))*)(uintptr_t)partial + (size_t)split * nq_pad;
#pragma unroll
    for (int r = 0; r < R; ++r) { const uint32_t qi = qbase + 64u * r; if (qi < (uint32_t)nq) prow[qi] = u32x2{ best[r], second[r] }; }
}

template <int R, int WAVES, int MODE>
float run(const u32x4* dQ, int nq, const u32x4* dT, int nt, int target_blocks, u32x2* dP, int reps, int* out_splits)
{
    const int qpb = 64 * R * WAVES;
    const int qblocks = (nq + qpb - 1) / qpb;
    int splits = std::max(1, (target_blocks + qblocks - 1) / qblocks);
    int per = (nt + splits - 1) / splits;
    splits = (nt + per - 1) / per;
    *out_splits = splits;
    const int nq_pad = (nq + 63) & ~63;
    if (splits > kMaxSplits || nq_pad > 10240) { printf("skip: %d splits exceed the partial buffer\n", splits); return 0.f; }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (MODE == 3) hipLaunchKernelGGL((sweep_lds<R, WAVES, 64>), dim3(qblocks * splits), dim3(64 * WAVES), 0, 0, dQ, nq, dT, nt, splits, per, dP, nq_pad);
        else if (MODE == 7) hipLaunchKernelGGL((sweep_wlds<R, WAVES, 1>), dim3(qblocks * splits), dim3(64 * WAVES), 0, 0, dQ, nq, dT, nt, splits, per, dP, nq_pad);
        else if (MODE == 8) hipLaunchKernelGGL((sweep_wlds<R, WAVES, 2>), dim3(qblocks * splits), dim3(64 * WAVES), 0, 0, dQ, nq, dT, nt, splits, per, dP, nq_pad);
        else if (MODE == 6) hipLaunchKernelGGL((sweep_wlds<R, WAVES>), dim3(qblocks * splits), dim3(64 * WAVES), 0, 0, dQ, nq, dT, nt, splits, per, dP, nq_pad);
        else if (MODE == 5) hipLaunchKernelGGL((sweep_lds<R, WAVES, 64, true>), dim3(qblocks * splits), dim3(64 * WAVES), 0, 0, dQ, nq, dT, nt, splits, per, dP, nq_pad);
        else if (MODE == 4) hipLaunchKernelGGL((sweep_lds<R, WAVES, 32>), dim3(qblocks * splits), dim3(64 * WAVES), 0, 0, dQ, nq, dT, nt, splits, per, dP, nq_pad);
        else hipLaunchKernelGGL((sweep<R, WAVES, (MODE == 9 || MODE == 10) ? MODE : (MODE > 2 ? 0 : MODE)>), dim3(qblocks * splits), dim3(64 * WAVES), 0, 0, dQ, nq, dT, nt, splits, per, dP, nq_pad);
    };
    for (int i = 0; i < 3; ++i) launch();
    CHECK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    {   // order-free checksum of (best, second) over splits: variants with equal split counts must agree
        std::vector<u32x2> hp((size_t)splits * nq_pad);
        CHECK(hipMemcpy(hp.data(), dP, hp.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long cs = 0;
        for (int sp = 0; sp < splits; ++sp) for (int qq = 0; qq < nq; ++qq) { const u32x2 e = hp[(size_t)sp * nq_pad + qq]; cs += (unsigned long long)e.x * 2654435761u + e.y; }
        g_checksum = cs;
    }
    return ts[ts.size() / 2];
}

int main(int argc, char** argv)
{
    const int nq = 10000, nt = 10000;
    std::vector<uint32_t> hq((size_t)nq * 16), ht((size_t)nt * 16);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (auto& v : hq) v = rnd();
    for (auto& v : ht) v = rnd();
    u32x4 *dQ, *dT; u32x2* dP;
    CHECK(hipMalloc((void**)&dQ, hq.size() * 4)); CHECK(hipMalloc((void**)&dT, ht.size() * 4 + 4096));
    CHECK(hipMalloc((void**)&dP, (size_t)kMaxSplits * 10240 * 8));
    CHECK(hipMemcpy(dQ, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dT, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    const double cmp = (double)nq * nt;
    const int targets[] = { 2048, 4096 };
    printf("%-28s %8s %7s %9s %9s\n", "variant", "target", "splits", "us", "Gcmp/s");
#define RUN(R, WV, MODE, name) for (int tb : targets) { int sp; float ms = run<R, WV, MODE>(dQ, nq, dT, nt, tb, dP, 15, &sp); \
        printf("%-28s %8d %7d %9.1f %9.1f  cs %016llx\n", name, tb, sp, ms * 1e3, cmp / (ms * 1e-3) / 1e9, g_checksum); }
    RUN(2, 4, 2, "R2 W4 prefetch (sgpr)");
    RUN(2, 4, 9, "R2 W4 prefetch sgpr paired-asm");
    RUN(2, 4, 10, "R2 W4 prefetch sgpr paired interleaved");
    RUN(1, 4, 9, "R1 W4 prefetch sgpr paired-asm");
    RUN(3, 4, 9, "R3 W4 prefetch sgpr paired-asm");
    RUN(4, 4, 9, "R4 W4 prefetch sgpr paired-asm");
    RUN(4, 4, 10, "R4 W4 prefetch sgpr paired interleaved");
    RUN(2, 8, 9, "R2 W8 prefetch sgpr paired-asm");
    RUN(2, 4, 7, "R2 W4 wave-lds paired interleaved");
    return 0;
}
