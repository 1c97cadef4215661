// capi.hip -- the C ABI of libcoloc_hip.so (include/coloc_hip.h): context, buffers, status codes.
//
// Host-side replacement for the CUDA-runtime plumbing of the reference's
// include/coloc/GPUDetector.hpp (ctor :70-138, detectAndDescribe :216-291) and
// include/coloc/GPUMatcher.hpp (ctor :70-95, setMapData :110-117, computeMatches :180-226,
// matchFeaturesWithMap :252-271).  No textures, no per-call object creation (the reference leaks a
// texture object per frame / per match call, GPUDetector.hpp:240,244, GPUMatcher.hpp:198-201), one
// stream per context, every HIP status checked and mapped to an int code.
#include "clc_internal.h"
#include "../host/HIPCovIntersection.hpp"
#include "../host/HIPRobustMatcher.hpp"      // hipgeom::motion_from_essential (host arithmetic of the inter-camera step)
#include "clc_acr.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace clc;

namespace clc {
// Event pairs recorded around kernel launches; drained (with a stream sync) by clc_profile_read.
struct Profiler {
    bool on = false;
    unsigned mask = 0xFFFFFFFFu;   // which kernels are bracketed
    struct Pair { hipEvent_t a = nullptr, b = nullptr; int kernel = 0; hipStream_t stream = nullptr; bool open = false; };
    std::vector<Pair> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[CLC_KERNEL_COUNT] = {};
    int launches[CLC_KERNEL_COUNT] = {};
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        // timing events need no system-scope release: a default event makes the GPU write its L2 back at every record, inside
        // the region being timed (the host never reads device data through these events, only their timestamps)
        if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) (void)hipEventCreate(&e);
        return e;
    }
    void drain()
    {
        for (Pair& p : pending) {
            if (!p.a || !p.b || p.open) continue;
            (void)hipEventSynchronize(p.b);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { total_ms[p.kernel] += ms; launches[p.kernel] += 1; }
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        pending.clear();
    }
    ~Profiler()
    {
        for (Pair& p : pending) { if (p.a) (void)hipEventDestroy(p.a); if (p.b) (void)hipEventDestroy(p.b); }
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    }
};
void prof_mark(Profiler* prof, int kernel, bool begin, hipStream_t stream)
{
    if (!prof || !prof->on || !((prof->mask >> kernel) & 1u)) return;
    if (begin) {
        Profiler::Pair p;
        p.a = prof->get(); p.b = prof->get(); p.kernel = kernel; p.stream = stream; p.open = true;
        if (p.a) (void)hipEventRecord(p.a, stream);
        prof->pending.push_back(p);
    } else {
        for (size_t i = prof->pending.size(); i-- > 0;) {
            Profiler::Pair& p = prof->pending[i];
            if (p.open && p.kernel == kernel && p.stream == stream) {
                if (p.b) (void)hipEventRecord(p.b, stream);
                p.open = false;
                break;
            }
        }
    }
}
} // namespace clc

struct clc_ctx {
    int device = 0;
    std::vector<float> acr_lg;   // (float) log10(k), k = 0 .. : the a-contrario tables are sums over it
    hipStream_t stream = nullptr;
    std::string err;
    bool has_det = false, has_mat = false;
    clc_detector_opts dopts{};
    clc_matcher_opts mopts{};
    // pyramid
    PyramidDesc pd{};
    uint8_t* d_arena = nullptr;
    size_t arena_bytes = 0;      // one pyramid
    int arena_slots = 1;         // pyramids the arena holds (grown by clc_describe_batch_dev)
    bool pyramid_valid = false;
    // detect + describe
    clc_keypoint* d_kps = nullptr;
    uint64_t* d_desc = nullptr;
    uint8_t* d_score = nullptr;      // arena-shaped FAST score maps (one per pyramid slot; only keypoint pixels are written and read)
    uint64_t* d_kpmask = nullptr;    // [slot][tile][16] keypoint bits of a tile row (detect.hip)
    uint32_t* d_tcount = nullptr;    // [slot][tile] keypoints of a tile
    uint32_t* d_count = nullptr;     // {written, found} of the context's own keypoint list
    uint32_t n_tiles = 0;
    bool detected = false;
    // match
    uint8_t* d_q = nullptr;
    uint8_t* d_t = nullptr;
    uint8_t* d_m = nullptr;
    int map_n = -1;
    int32_t* d_match = nullptr;
    uint16_t* d_best = nullptr;
    uint16_t* d_second = nullptr;
    uint2* d_partial = nullptr;
    size_t partial_cap = 0;
    bool partial_dirty = false;      // armed (all-ones) state of the atomic top-2 rows was lost
    int formulation = K2NN_MATRIX;   // K2NN sweep formulation (k2nn.hip): FP4 matrix pipe, or round 1's popcount kernel for A/B runs
    int target_blocks = 0;           // K2NN sweep workgroups aimed at per launch; 0 = the formulation's default
    int bias_a = 326, bias_b = 249;  // matrix sweep, one-round single-job plans: train share of a workgroup on wave slot 0 / 1 in 1/256 of the
                                     // equal share (k2nn.hip; measured optimum 21 : 16 : 12-13 tiles at 10k x 10k); CLC_K2NN_BIAS=a,b, 0,0 = equal shares
    bool xcd_map = true;         // XCD-aware K2NN tile order (CLC_K2NN_XCD_MAP=0 switches it off for A/B runs)
    K2nnDevice k2dev{};          // XCDs and CUs of this context's device (the sweep planner's balance arguments)
    int bias_source = 0;         // 0: built-in default, 1: CLC_K2NN_BIAS, 2: timed probe on this device (k2nn_probe_bias)
    float bias_probe_us[4] = {}; // the probe's sweep times per candidate (0: not probed)
    hipEvent_t ev_group = nullptr;   // drive_group: the tail of a batch's shared launches, for the other contexts' streams to wait on
    int cache_mode = CLC_DESC_CACHE_VERIFY;   // how this context's host-pointer match entry points treat published blocks (clc_desc_cache_mode)
    // pnp
    uint8_t* d_pairs = nullptr;   // clc_match_pairs arena: descriptors of all cameras, then results
    size_t pairs_cap = 0;
    double* d_pnp = nullptr;
    size_t pnp_cap = 0;   // doubles
    void* h_pin = nullptr;        // pinned staging for the pose solve
    size_t pin_cap = 0;
    Profiler prof;
};

namespace {

int fail(clc_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess)
{
    if (ctx) {
        char buf[512];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
        else snprintf(buf, sizeof buf, "%s", what);
        ctx->err = buf;
        // Any HIP failure on this context -- including one that only surfaces at a later synchronisation -- may have cut a
        // K2NN sweep short and left top-2 rows / arrival counters un-armed: re-arm the workspace before the next sweep.
        if (code == CLC_ERR_HIP) ctx->partial_dirty = true;
    }
    return code;
}

#define CLC_HIP(ctx, call)                                                      \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) return fail((ctx), CLC_ERR_HIP, #call, e__);     \
    } while (0)

hipStream_t pick(clc_ctx* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }

uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

// GPUDetector.hpp:109-114: f_i = f_{i-1} * scale_factor, dims (uint32)((float)W / f + 0.5f)
void plan_pyramid(const clc_detector_opts& o, PyramidDesc& pd, size_t& bytes)
{
    pd.levels = o.scale_levels;
    float f = 1.0f;
    uint32_t off = 0, blk = 0;
    for (int i = 0; i < pd.levels; ++i) {
        if (i) f *= o.scale_factor;
        const uint32_t w = i ? (uint32_t)((float)o.width / f + 0.5f) : o.width;
        const uint32_t h = i ? (uint32_t)((float)o.height / f + 0.5f) : o.height;
        pd.f[i] = f;
        pd.lv[i].w = w;
        pd.lv[i].h = h;
        pd.lv[i].pitch = align_up(w ? w : 1, 64);
        pd.lv[i].offset = off;
        off += align_up(pd.lv[i].pitch * (h ? h : 1), 256);
        pd.blk_begin[i] = blk;
        if (i) blk += (pd.lv[i].pitch / 4 * h + 255) / 256;
    }
    for (int i = pd.levels; i <= CLC_MAX_LEVELS; ++i) pd.blk_begin[i] = blk;
    bytes = off + 256;
}

int ensure_partial(clc_ctx* ctx, size_t elems)
{
    if (elems <= ctx->partial_cap) return CLC_OK;
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_partial) CLC_HIP(ctx, hipFree(ctx->d_partial));
    ctx->d_partial = nullptr;
    ctx->partial_cap = 0;
    size_t cap = elems + elems / 4;
    CLC_HIP(ctx, hipMalloc((void**)&ctx->d_partial, cap * sizeof(uint2)));
    CLC_HIP(ctx, hipMemsetAsync(ctx->d_partial, 0xFF, cap * sizeof(uint2), ctx->stream));   // armed top-2 rows
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->partial_cap = cap;
    return CLC_OK;
}

int ensure_pnp(clc_ctx* ctx, size_t doubles)
{
    if (doubles <= ctx->pnp_cap) return CLC_OK;
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_pnp) CLC_HIP(ctx, hipFree(ctx->d_pnp));
    ctx->d_pnp = nullptr;
    ctx->pnp_cap = 0;
    // half as much again: a stream of solves whose sizes creep upwards (map matches per frame) must not reallocate at every new maximum
    const size_t cap = doubles + doubles / 2;
    CLC_HIP(ctx, hipMalloc((void**)&ctx->d_pnp, cap * sizeof(double)));
    ctx->pnp_cap = cap;
    return CLC_OK;
}

int cache_mode_default()
{
    const char* v = getenv("CLC_DESC_CACHE");
    if (!v || !v[0]) return CLC_DESC_CACHE_VERIFY;
    if (v[0] == '0' || v[0] == 'o' || v[0] == 'O') return CLC_DESC_CACHE_OFF;
    if (v[0] == 't' || v[0] == 'T' || v[0] == '2') return CLC_DESC_CACHE_TRUST;
    return CLC_DESC_CACHE_VERIFY;
}

// matrix: 3 workgroups of 4 waves per CU (152 VGPRs) = ONE resident round; popcount: 10 workgroups of 8 waves per CU (measured optima);
// per CU of the context's device (768 / 2560 on MI355X's 256)
int default_target_blocks(int formulation, const K2nnDevice& dev = K2nnDevice{}) { return (int)((formulation == K2NN_POPCOUNT ? 10u : 3u) * dev.n_cu); }

int run_jobs(clc_ctx* ctx, std::vector<K2nnJobDev>& jobs, hipStream_t st, bool probe = false)
{
    const int target = ctx->target_blocks > 0 ? ctx->target_blocks : default_target_blocks(ctx->formulation, ctx->k2dev);
    const K2nnPlan plan = k2nn_plan(jobs.data(), (int)jobs.size(), target, ctx->xcd_map, ctx->formulation, ctx->bias_a, ctx->bias_b, ctx->k2dev);
    const int rc = ensure_partial(ctx, plan.partial_elems);
    if (rc != CLC_OK) return rc;
    if (!plan.atomic_merge) ctx->partial_dirty = true;           // slab mode scribbles over the armed rows
    else if (ctx->partial_dirty) {
        CLC_HIP(ctx, hipMemsetAsync(ctx->d_partial, 0xFF, ctx->partial_cap * sizeof(uint2), st));
        ctx->partial_dirty = false;
    }
    const hipError_t e = launch_k2nn(jobs.data(), (int)jobs.size(), ctx->d_partial, st, probe ? nullptr : &ctx->prof, ctx->formulation, nullptr, probe);
    if (e != hipSuccess) { ctx->partial_dirty = true; return fail(ctx, CLC_ERR_HIP, "launch_k2nn", e); }
    return CLC_OK;
}

// ---- which unequal shares suit THIS device (round 5) ---------------------------------------------------------------------------------
// The share a slot-0 / slot-1 workgroup takes of a one-round sweep has an optimum that moves from device to device (19 : 17 ... 21 : 15
// tiles of a 313-tile train set across the MI355X boxes of rounds 4-5: the chips hold different clocks under the sweep's load).  The first
// matcher context a process creates on a device times the 10k x 10k sweep under four candidate pairs -- behind 4 ms of the same sweep, so
// that the clocks are up -- and every later context on that device takes the winner.  ~15 ms once per process and device; results do not
// depend on it (the fold is order-free).  CLC_K2NN_BIAS=a,b or CLC_K2NN_PROBE=0 skip it.
__global__ void k2nn_probe_fill_kernel(uint32_t* p, const size_t n, const uint32_t salt)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u + salt;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    p[i] = x;
}
struct K2nnProbeCache { std::mutex mu; bool done[64] = {}; int a[64] = {}, b[64] = {}; float us[64][4] = {}; };
static K2nnProbeCache& probe_cache() { static K2nnProbeCache c; return c; }
static constexpr int kProbeCand[4][2] = { { 295, 264 }, { 311, 256 }, { 326, 249 }, { 326, 233 } };     // 19:17, 20:16.5, 21:16 (rounds 4's default), 21:15

static void k2nn_probe_bias(clc_ctx* ctx)
{
    const int N = 10000;
    if (!ctx->has_mat || ctx->mopts.maxkp < (uint32_t)N || ctx->formulation == K2NN_POPCOUNT || ctx->device < 0 || ctx->device >= 64) return;
    if (ctx->k2dev.n_xcd != kK2nnXcds || ctx->target_blocks > 0) return;
    if (const char* e = getenv("CLC_K2NN_PROBE")) if (e[0] == '0') return;
    K2nnProbeCache& pc = probe_cache();
    std::lock_guard<std::mutex> lk(pc.mu);
    const int d = ctx->device;
    if (!pc.done[d]) {
        pc.done[d] = true; pc.a[d] = ctx->bias_a; pc.b[d] = ctx->bias_b;             // whatever happens below: probe once
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) (void)hipEventDestroy(e0); return; }
        const size_t words = (size_t)N * CLC_DESC_BYTES / 4;
        hipLaunchKernelGGL(k2nn_probe_fill_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)ctx->d_q, words, 1u);
        hipLaunchKernelGGL(k2nn_probe_fill_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)ctx->d_t, words, 2u);
        std::vector<K2nnJobDev> jobs(1);
        auto sweep = [&](const int a, const int b, const int reps) -> bool {
            const int sa = ctx->bias_a, sb = ctx->bias_b;
            ctx->bias_a = a; ctx->bias_b = b;
            bool ok = true;
            for (int r = 0; r < reps && ok; ++r) {
                jobs[0] = K2nnJobDev{};
                jobs[0].q = (const uint4*)ctx->d_q; jobs[0].t = (const uint4*)ctx->d_t; jobs[0].out = ctx->d_match;
                jobs[0].nq = (uint32_t)N; jobs[0].nt = (uint32_t)N; jobs[0].thr = 40u;
                ok = run_jobs(ctx, jobs, ctx->stream, true) == CLC_OK;
            }
            ctx->bias_a = sa; ctx->bias_b = sb;
            return ok;
        };
        bool ok = sweep(ctx->bias_a, ctx->bias_b, 160);                                  // ~4 ms: clocks up
        float best = 0.f; int best_i = -1;
        for (int pass = 0; pass < 2 && ok; ++pass)                                        // two interleaved passes, the smaller time counts
            for (int c = 0; c < 4 && ok; ++c) {
                ok = hipEventRecord(e0, ctx->stream) == hipSuccess && sweep(kProbeCand[c][0], kProbeCand[c][1], 40) && hipEventRecord(e1, ctx->stream) == hipSuccess &&
                     hipEventSynchronize(e1) == hipSuccess;
                float ms = 0.f;
                if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
                    const float us = ms * 1000.f / 40.f;
                    if (pc.us[d][c] == 0.f || us < pc.us[d][c]) pc.us[d][c] = us;
                }
            }
        if (ok)
            for (int c = 0; c < 4; ++c) if (pc.us[d][c] > 0.f && (best_i < 0 || pc.us[d][c] < best)) { best = pc.us[d][c]; best_i = c; }
        // the winner must beat rounds 4's default by more than the noise of such a short measurement (1 %), else the default stays
        if (ok && best_i >= 0 && pc.us[d][2] > 0.f && best < 0.99f * pc.us[d][2]) { pc.a[d] = kProbeCand[best_i][0]; pc.b[d] = kProbeCand[best_i][1]; }
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (!ok) { ctx->partial_dirty = true; for (int c = 0; c < 4; ++c) pc.us[d][c] = 0.f; }
    }
    ctx->bias_a = pc.a[d]; ctx->bias_b = pc.b[d];
    for (int c = 0; c < 4; ++c) ctx->bias_probe_us[c] = pc.us[d][c];
    if (pc.us[d][0] > 0.f) ctx->bias_source = 2;
}

} // namespace

extern "C" {

int clc_abi_version(void) { return CLC_ABI_VERSION; }

const char* clc_status_string(int status)
{
    switch (status) {
        case CLC_OK: return "ok";
        case CLC_ERR_BAD_ARG: return "bad argument";
        case CLC_ERR_CAPACITY: return "capacity exceeded";
        case CLC_ERR_HIP: return "HIP runtime error";
        case CLC_ERR_NO_DEVICE: return "no usable device";
        case CLC_ERR_STATE: return "invalid call order";
        default: return "unknown status";
    }
}

const char* clc_last_error_string(const clc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int clc_ctx_create(int device_id, const clc_detector_opts* dopts, const clc_matcher_opts* mopts, clc_ctx** out_ctx)
{
    if (!out_ctx) return CLC_ERR_BAD_ARG;
    *out_ctx = nullptr;
    if (dopts) {
        if (dopts->scale_levels < 1 || dopts->scale_levels > CLC_MAX_LEVELS || dopts->width < 8 || dopts->height < 8 ||
            !(dopts->scale_factor > 1.0f) || dopts->maxkp == 0)
            return CLC_ERR_BAD_ARG;
        if (dopts->width > CLC_DETECT_MAX_WIDTH) {
            // (said here, with its own code: every later clc_detect* call would otherwise fail with a bare hipErrorInvalidValue; there is no
            // context yet whose clc_last_error_string could carry the text)
            fprintf(stderr, "coloc_hip: DetectorOptions.width %u exceeds CLC_DETECT_MAX_WIDTH %d (the GPU detector keeps a row's pre-test bits in LDS)\n",
                    dopts->width, CLC_DETECT_MAX_WIDTH);
            return CLC_ERR_CAPACITY;
        }
    }
    if (mopts && mopts->maxkp == 0) return CLC_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return CLC_ERR_NO_DEVICE;
    clc_ctx* ctx = new (std::nothrow) clc_ctx;
    if (!ctx) return CLC_ERR_HIP;
    ctx->device = device_id;
#define CREATE_HIP(call)                                     \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) {                             \
            fprintf(stderr, "coloc_hip: %s failed: %s\n", #call, hipGetErrorString(e__)); \
            clc_ctx_destroy(ctx);                            \
            return CLC_ERR_HIP;                              \
        }                                                    \
    } while (0)
    CREATE_HIP(hipSetDevice(device_id));
    CREATE_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && v > 0) ctx->k2dev.n_cu = (uint32_t)v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, device_id) == hipSuccess && v > 0) ctx->k2dev.n_xcd = (uint32_t)v;
    }
    if (const char* e = getenv("CLC_K2NN_TARGET_BLOCKS")) {
        const int v = atoi(e);
        if (v > 0) ctx->target_blocks = v;
    }
    if (const char* e = getenv("CLC_K2NN_XCD_MAP")) ctx->xcd_map = atoi(e) != 0;
    ctx->cache_mode = cache_mode_default();
    if (const char* e = getenv("CLC_K2NN_BIAS")) {
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && a >= 0 && b >= 0 && a < 1024 && b < 1024) { ctx->bias_a = a; ctx->bias_b = b; ctx->bias_source = 1; }
    }
    if (const char* e = getenv("CLC_K2NN_FORMULATION"))
        ctx->formulation = (e[0] == 'p' || e[0] == '1') ? K2NN_POPCOUNT : K2NN_MATRIX;
    if (dopts) {
        ctx->has_det = true;
        ctx->dopts = *dopts;
        plan_pyramid(*dopts, ctx->pd, ctx->arena_bytes);
        CREATE_HIP(hipMalloc((void**)&ctx->d_arena, ctx->arena_bytes));
        CREATE_HIP(hipMemsetAsync(ctx->d_arena, 0, ctx->arena_bytes, ctx->stream));
        CREATE_HIP(hipMalloc((void**)&ctx->d_kps, (size_t)dopts->maxkp * sizeof(clc_keypoint)));
        CREATE_HIP(hipMalloc((void**)&ctx->d_desc, (size_t)dopts->maxkp * CLC_DESC_BYTES));
        CREATE_HIP(hipMalloc((void**)&ctx->d_score, ctx->arena_bytes));
        ctx->n_tiles = detect_total_tiles(ctx->pd);
        CREATE_HIP(hipMalloc((void**)&ctx->d_kpmask, (size_t)ctx->n_tiles * 16 * sizeof(uint64_t)));
        CREATE_HIP(hipMalloc((void**)&ctx->d_tcount, (size_t)ctx->n_tiles * sizeof(uint32_t)));
        CREATE_HIP(hipMalloc((void**)&ctx->d_count, 4 * sizeof(uint32_t)));
        CREATE_HIP(hipMemsetAsync(ctx->d_count, 0, 4 * sizeof(uint32_t), ctx->stream));
    }
    if (mopts) {
        ctx->has_mat = true;
        ctx->mopts = *mopts;
        const size_t cap = (size_t)mopts->maxkp;
        CREATE_HIP(hipMalloc((void**)&ctx->d_q, cap * CLC_DESC_BYTES));
        CREATE_HIP(hipMalloc((void**)&ctx->d_t, cap * CLC_DESC_BYTES));
        CREATE_HIP(hipMalloc((void**)&ctx->d_m, cap * CLC_DESC_BYTES));
        CREATE_HIP(hipMalloc((void**)&ctx->d_match, cap * sizeof(int32_t)));
        CREATE_HIP(hipMalloc((void**)&ctx->d_best, cap * sizeof(uint16_t)));
        CREATE_HIP(hipMalloc((void**)&ctx->d_second, cap * sizeof(uint16_t)));
    }
    // K2NN workspace: one armed {best, second} row per query suffices in atomic mode (a few pairs' worth
    // here); the slab fallback for train sets > 2^22 and larger job lists grow it on demand
    {
        const size_t cap = dopts || mopts ? (size_t)(mopts ? mopts->maxkp : dopts->maxkp) : 16384;
        const size_t elems = ((cap + 63) & ~(size_t)63) * 8 + 4096;
        CREATE_HIP(hipMalloc((void**)&ctx->d_partial, elems * sizeof(uint2)));
        CREATE_HIP(hipMemsetAsync(ctx->d_partial, 0xFF, elems * sizeof(uint2), ctx->stream));   // armed top-2 rows
        ctx->partial_cap = elems;
    }
    CREATE_HIP(hipStreamSynchronize(ctx->stream));
#undef CREATE_HIP
    if (ctx->bias_source == 0) k2nn_probe_bias(ctx);
    *out_ctx = ctx;
    return CLC_OK;
}

int clc_ctx_destroy(clc_ctx* ctx)
{
    if (!ctx) return CLC_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    void* bufs[] = { ctx->d_arena, ctx->d_kps, ctx->d_desc, ctx->d_score, ctx->d_kpmask, ctx->d_tcount, ctx->d_count, ctx->d_q, ctx->d_t, ctx->d_m, ctx->d_match,
                     ctx->d_best, ctx->d_second, ctx->d_partial, ctx->d_pnp, ctx->d_pairs };
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->ev_group) (void)hipEventDestroy(ctx->ev_group);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CLC_OK;
}

int clc_sync(clc_ctx* ctx)
{
    if (!ctx) return CLC_ERR_BAD_ARG;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

void* clc_stream(clc_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int clc_ctx_device(const clc_ctx* ctx) { return ctx ? ctx->device : -1; }

int clc_profile_enable(clc_ctx* ctx, int on)
{
    if (!ctx) return CLC_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    if (!on) ctx->prof.drain();
    ctx->prof.on = on != 0;
    // on = 1: every kernel; otherwise bit (k + 1) selects kernel k, e.g. 1 << (CLC_KERNEL_K2NN_SWEEP + 1)
    ctx->prof.mask = (on == 1 || on == 0) ? 0xFFFFFFFFu : ((unsigned)on >> 1);
    return CLC_OK;
}

int clc_profile_reset(clc_ctx* ctx)
{
    if (!ctx) return CLC_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    ctx->prof.drain();
    for (int k = 0; k < CLC_KERNEL_COUNT; ++k) { ctx->prof.total_ms[k] = 0.0; ctx->prof.launches[k] = 0; }
    return CLC_OK;
}

int clc_profile_read(clc_ctx* ctx, int kernel, double* total_ms, int* launches)
{
    if (!ctx || kernel < 0 || kernel >= CLC_KERNEL_COUNT) return CLC_ERR_BAD_ARG;
    (void)hipSetDevice(ctx->device);
    ctx->prof.drain();
    if (total_ms) *total_ms = ctx->prof.total_ms[kernel];
    if (launches) *launches = ctx->prof.launches[kernel];
    return CLC_OK;
}

const char* clc_kernel_name(int kernel)
{
    static const char* names[CLC_KERNEL_COUNT] = { "pyramid_kernel", "clatch_kernel", "k2nn_sweep_kernel",
                                                    "k2nn_merge_kernel", "pnp_residual_kernel", "pnp_score_kernel",
                                                    "detect_kernels" };
    return (kernel >= 0 && kernel < CLC_KERNEL_COUNT) ? names[kernel] : "?";
}

/* ---- pyramid ------------------------------------------------------------------------------- */

int clc_pyramid_build_dev(clc_ctx* ctx, const void* d_img, uint32_t width, uint32_t height, size_t pitch, void* stream)
{
    if (!ctx || !d_img) return fail(ctx, CLC_ERR_BAD_ARG, "pyramid_build: null argument");
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "pyramid_build: context created without detector options");
    if (width != ctx->dopts.width || height != ctx->dopts.height || pitch < width)
        return fail(ctx, CLC_ERR_BAD_ARG, "pyramid_build: image size differs from DetectorOptions width/height");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    if (pitch > 0xFFFFFFFFull) return fail(ctx, CLC_ERR_BAD_ARG, "pyramid_build: pitch too large");
    CLC_HIP(ctx, launch_pyramid(ctx->pd, ctx->d_arena, (const uint8_t*)d_img, (uint32_t)pitch, st, &ctx->prof));
    ctx->pyramid_valid = true;
    return CLC_OK;
}

int clc_pyramid_build(clc_ctx* ctx, const uint8_t* h_img, uint32_t width, uint32_t height)
{
    if (!ctx || !h_img) return fail(ctx, CLC_ERR_BAD_ARG, "pyramid_build: null argument");
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "pyramid_build: context created without detector options");
    if (width != ctx->dopts.width || height != ctx->dopts.height)
        return fail(ctx, CLC_ERR_BAD_ARG, "pyramid_build: image size differs from DetectorOptions width/height");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    const LevelDesc& L0 = ctx->pd.lv[0];
    CLC_HIP(ctx, hipMemcpy2DAsync(ctx->d_arena + L0.offset, L0.pitch, h_img, width, width, height,
                                  hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, launch_pyramid(ctx->pd, ctx->d_arena, ctx->d_arena + L0.offset, L0.pitch, ctx->stream, &ctx->prof));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pyramid_valid = true;
    return CLC_OK;
}

int clc_pyramid_level(const clc_ctx* ctx, int level, uint32_t* w, uint32_t* h, size_t* pitch, const void** d_ptr)
{
    if (!ctx || !ctx->has_det || level < 0 || level >= ctx->pd.levels) return CLC_ERR_BAD_ARG;
    const LevelDesc& L = ctx->pd.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (pitch) *pitch = L.pitch;
    if (d_ptr) *d_ptr = ctx->d_arena + L.offset;
    return CLC_OK;
}

int clc_pyramid_download(clc_ctx* ctx, int level, uint8_t* h_out)
{
    if (!ctx || !h_out) return fail(ctx, CLC_ERR_BAD_ARG, "pyramid_download: null argument");
    if (!ctx->has_det || level < 0 || level >= ctx->pd.levels) return fail(ctx, CLC_ERR_BAD_ARG, "pyramid_download: bad level");
    if (!ctx->pyramid_valid) return fail(ctx, CLC_ERR_STATE, "pyramid_download before pyramid_build");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    const LevelDesc& L = ctx->pd.lv[level];
    CLC_HIP(ctx, hipMemcpy2DAsync(h_out, L.w, ctx->d_arena + L.offset, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

/* ---- detect -------------------------------------------------------------------------------- */

static uint32_t* count_ptr(clc_ctx* ctx) { return ctx->d_count; }

// room for the pyramids (+ score maps, keypoint masks, tile counts) of n cameras; slot 0 (the current single-image pyramid) does not
// survive a growth -- every caller rebuilds it
static int ensure_slots(clc_ctx* ctx, int n, hipStream_t st)
{
    if (n <= ctx->arena_slots) return CLC_OK;
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st != ctx->stream) CLC_HIP(ctx, hipStreamSynchronize(st));
    uint8_t *arena = nullptr, *score = nullptr;
    uint64_t* kpmask = nullptr;
    uint32_t* tcount = nullptr;
    hipError_t e = hipMalloc((void**)&arena, ctx->arena_bytes * (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void**)&score, ctx->arena_bytes * (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void**)&kpmask, (size_t)n * ctx->n_tiles * 16 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void**)&tcount, (size_t)n * ctx->n_tiles * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(arena, 0, ctx->arena_bytes * (size_t)n, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        (void)hipFree(arena); (void)hipFree(score); (void)hipFree(kpmask); (void)hipFree(tcount);
        return fail(ctx, CLC_ERR_HIP, "growing the pyramid arena", e);
    }
    ctx->pyramid_valid = false;
    ctx->detected = false;
    (void)hipFree(ctx->d_arena); (void)hipFree(ctx->d_score); (void)hipFree(ctx->d_kpmask); (void)hipFree(ctx->d_tcount);
    ctx->d_arena = arena; ctx->d_score = score; ctx->d_kpmask = kpmask; ctx->d_tcount = tcount;
    ctx->arena_slots = n;
    return CLC_OK;
}

int clc_detect_dev(clc_ctx* ctx, void* stream)
{
    if (!ctx) return CLC_ERR_BAD_ARG;
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "detect: context created without detector options");
    if (!ctx->pyramid_valid) return fail(ctx, CLC_ERR_STATE, "detect before pyramid_build");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    clc_keypoint* kps[1] = { ctx->d_kps };
    uint32_t* cnt[1] = { ctx->d_count };
    CLC_HIP(ctx, launch_detect(ctx->pd, ctx->d_arena, ctx->arena_bytes, 1, ctx->d_score, ctx->d_kpmask, ctx->d_tcount, ctx->dopts.thresh,
                               ctx->dopts.maxkp, kps, cnt, pick(ctx, stream), &ctx->prof));
    ctx->detected = true;
    return CLC_OK;
}

int clc_detect_batch_dev(clc_ctx* ctx, int n_images, const void* const* d_imgs, uint32_t width, uint32_t height, size_t pitch,
                         clc_keypoint* const* d_kps, uint32_t* const* d_counts, void* const* d_desc, void* stream)
{
    if (!ctx || n_images < 0 || (n_images > 0 && (!d_imgs || !d_kps || !d_counts)))
        return fail(ctx, CLC_ERR_BAD_ARG, "detect_batch: null argument");
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "detect_batch: context created without detector options");
    if (n_images > CLC_MAX_BATCH) return fail(ctx, CLC_ERR_CAPACITY, "detect_batch: more than CLC_MAX_BATCH images");
    if (n_images == 0) return CLC_OK;
    if (width != ctx->dopts.width || height != ctx->dopts.height || pitch < width || pitch > 0xFFFFFFFFull)
        return fail(ctx, CLC_ERR_BAD_ARG, "detect_batch: image size differs from DetectorOptions width/height");
    ClatchBatch batch{};
    const uint8_t* srcs[CLC_MAX_BATCH] = {};
    const uint32_t* cnts[CLC_MAX_BATCH] = {};
    for (int b = 0; b < n_images; ++b) {
        if (!d_imgs[b] || !d_kps[b] || !d_counts[b] || (d_desc && !d_desc[b]))
            return fail(ctx, CLC_ERR_BAD_ARG, "detect_batch: null image / keypoint / count / descriptor pointer");
        if (((uintptr_t)d_kps[b] & 3u) || ((uintptr_t)d_counts[b] & 3u) || (d_desc && ((uintptr_t)d_desc[b] & 7u)))
            return fail(ctx, CLC_ERR_BAD_ARG, "detect_batch: misaligned device pointer");
        srcs[b] = (const uint8_t*)d_imgs[b];
        batch.kps[b] = d_kps[b];
        batch.desc[b] = d_desc ? (uint64_t*)d_desc[b] : nullptr;
        batch.n[b] = (int)ctx->dopts.maxkp;
        cnts[b] = d_counts[b];
    }
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    const int rc = ensure_slots(ctx, n_images, st);
    if (rc != CLC_OK) return rc;
    ctx->pyramid_valid = false;
    CLC_HIP(ctx, launch_pyramid_batch(ctx->pd, ctx->d_arena, ctx->arena_bytes, srcs, n_images, (uint32_t)pitch, st, &ctx->prof));
    ctx->pyramid_valid = true;
    ctx->detected = false;               // the context's own keypoint list is not the one this call fills
    CLC_HIP(ctx, launch_detect(ctx->pd, ctx->d_arena, ctx->arena_bytes, n_images, ctx->d_score, ctx->d_kpmask, ctx->d_tcount,
                               ctx->dopts.thresh, ctx->dopts.maxkp, d_kps, d_counts, st, &ctx->prof));
    if (d_desc)
        CLC_HIP(ctx, launch_clatch_counted_batch(ctx->pd, ctx->d_arena, ctx->arena_bytes, batch, cnts, n_images, st, &ctx->prof));
    return CLC_OK;
}

int clc_detect_buffers(clc_ctx* ctx, const clc_keypoint** d_kps, const uint32_t** d_count, void** d_desc)
{
    if (!ctx || !ctx->has_det) return CLC_ERR_BAD_ARG;
    if (d_kps) *d_kps = ctx->d_kps;
    if (d_count) *d_count = count_ptr(ctx);
    if (d_desc) *d_desc = ctx->d_desc;
    return CLC_OK;
}

int clc_detect(clc_ctx* ctx, clc_keypoint* h_kps, int capacity, int* n_written, int* n_found)
{
    if (!ctx || capacity < 0 || (capacity > 0 && !h_kps)) return fail(ctx, CLC_ERR_BAD_ARG, "detect: bad argument");
    const int rc = clc_detect_dev(ctx, nullptr);
    if (rc != CLC_OK) return rc;
    uint32_t cnt[2] = { 0, 0 };
    CLC_HIP(ctx, hipMemcpyAsync(cnt, count_ptr(ctx), sizeof cnt, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int n = (int)cnt[0] < capacity ? (int)cnt[0] : capacity;
    if (n > 0) {
        CLC_HIP(ctx, hipMemcpyAsync(h_kps, ctx->d_kps, (size_t)n * sizeof(clc_keypoint), hipMemcpyDeviceToHost, ctx->stream));
        CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n_written) *n_written = n;
    if (n_found) *n_found = (int)cnt[1];
    return CLC_OK;
}

int clc_describe_detected_dev(clc_ctx* ctx, void* d_desc, void* stream)
{
    if (!ctx) return CLC_ERR_BAD_ARG;
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "describe_detected: context created without detector options");
    if (!ctx->detected) return fail(ctx, CLC_ERR_STATE, "describe_detected before detect");
    if ((uintptr_t)d_desc & 7u) return fail(ctx, CLC_ERR_BAD_ARG, "describe_detected: misaligned device pointer");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    CLC_HIP(ctx, launch_clatch_counted(ctx->pd, ctx->d_arena, ctx->d_kps, count_ptr(ctx), (int)ctx->dopts.maxkp,
                                       d_desc ? (uint64_t*)d_desc : ctx->d_desc, pick(ctx, stream), &ctx->prof));
    return CLC_OK;
}

int clc_detect_and_describe(clc_ctx* ctx, const uint8_t* h_img, uint32_t width, uint32_t height, clc_keypoint* h_kps,
                            uint8_t* h_desc, int capacity, int* n_written, int* n_found)
{
    if (!ctx || !h_img || capacity < 0 || (capacity > 0 && (!h_kps || !h_desc)))
        return fail(ctx, CLC_ERR_BAD_ARG, "detect_and_describe: bad argument");
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "detect_and_describe: context created without detector options");
    if (width != ctx->dopts.width || height != ctx->dopts.height)
        return fail(ctx, CLC_ERR_BAD_ARG, "detect_and_describe: image size differs from DetectorOptions width/height");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    const LevelDesc& L0 = ctx->pd.lv[0];
    CLC_HIP(ctx, hipMemcpy2DAsync(ctx->d_arena + L0.offset, L0.pitch, h_img, width, width, height, hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, launch_pyramid(ctx->pd, ctx->d_arena, ctx->d_arena + L0.offset, L0.pitch, ctx->stream, &ctx->prof));
    ctx->pyramid_valid = true;
    int rc = clc_detect_dev(ctx, nullptr);
    if (rc != CLC_OK) return rc;
    rc = clc_describe_detected_dev(ctx, nullptr, nullptr);
    if (rc != CLC_OK) return rc;
    uint32_t cnt[2] = { 0, 0 };
    CLC_HIP(ctx, hipMemcpyAsync(cnt, count_ptr(ctx), sizeof cnt, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int n = (int)cnt[0] < capacity ? (int)cnt[0] : capacity;
    if (n > 0) {
        CLC_HIP(ctx, hipMemcpyAsync(h_kps, ctx->d_kps, (size_t)n * sizeof(clc_keypoint), hipMemcpyDeviceToHost, ctx->stream));
        CLC_HIP(ctx, hipMemcpyAsync(h_desc, ctx->d_desc, (size_t)n * CLC_DESC_BYTES, hipMemcpyDeviceToHost, ctx->stream));
        CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n_written) *n_written = n;
    if (n_found) *n_found = (int)cnt[1];
    return CLC_OK;
}

/* ---- describe ------------------------------------------------------------------------------ */

int clc_describe_dev(clc_ctx* ctx, const clc_keypoint* d_kps, int n, void* d_desc, void* stream)
{
    if (!ctx || n < 0 || (n > 0 && (!d_kps || !d_desc))) return fail(ctx, CLC_ERR_BAD_ARG, "describe: bad argument");
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "describe: context created without detector options");
    if (!ctx->pyramid_valid) return fail(ctx, CLC_ERR_STATE, "describe before pyramid_build");
    if (((uintptr_t)d_desc & 7u) || ((uintptr_t)d_kps & 3u)) return fail(ctx, CLC_ERR_BAD_ARG, "describe: misaligned device pointer");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    CLC_HIP(ctx, launch_clatch(ctx->pd, ctx->d_arena, d_kps, n, (uint64_t*)d_desc, pick(ctx, stream), &ctx->prof));
    return CLC_OK;
}

int clc_describe_batch_dev(clc_ctx* ctx, int n_images, const void* const* d_imgs, uint32_t width, uint32_t height,
                           size_t pitch, const clc_keypoint* const* d_kps, const int* counts, void* const* d_desc,
                           void* stream)
{
    if (!ctx || n_images < 0 || (n_images > 0 && (!d_imgs || !d_kps || !counts || !d_desc)))
        return fail(ctx, CLC_ERR_BAD_ARG, "describe_batch: null argument");
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "describe_batch: context created without detector options");
    if (n_images > CLC_MAX_BATCH) return fail(ctx, CLC_ERR_CAPACITY, "describe_batch: more than CLC_MAX_BATCH images");
    if (n_images == 0) return CLC_OK;
    if (width != ctx->dopts.width || height != ctx->dopts.height || pitch < width || pitch > 0xFFFFFFFFull)
        return fail(ctx, CLC_ERR_BAD_ARG, "describe_batch: image size differs from DetectorOptions width/height");
    ClatchBatch batch{};
    const uint8_t* srcs[CLC_MAX_BATCH] = {};
    for (int b = 0; b < n_images; ++b) {
        if (!d_imgs[b] || counts[b] < 0 || (counts[b] > 0 && (!d_kps[b] || !d_desc[b])))
            return fail(ctx, CLC_ERR_BAD_ARG, "describe_batch: null image / keypoint / descriptor pointer");
        if (((uintptr_t)d_desc[b] & 7u) || ((uintptr_t)d_kps[b] & 3u))
            return fail(ctx, CLC_ERR_BAD_ARG, "describe_batch: misaligned device pointer");
        srcs[b] = (const uint8_t*)d_imgs[b];
        batch.kps[b] = d_kps[b];
        batch.desc[b] = (uint64_t*)d_desc[b];
        batch.n[b] = counts[b];
    }
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    {
        const int rc = ensure_slots(ctx, n_images, st);
        if (rc != CLC_OK) return rc;
    }
    ctx->pyramid_valid = false;
    CLC_HIP(ctx, launch_pyramid_batch(ctx->pd, ctx->d_arena, ctx->arena_bytes, srcs, n_images, (uint32_t)pitch, st, &ctx->prof));
    ctx->pyramid_valid = true;
    CLC_HIP(ctx, launch_clatch_batch(ctx->pd, ctx->d_arena, ctx->arena_bytes, batch, n_images, st, &ctx->prof));
    return CLC_OK;
}

int clc_describe(clc_ctx* ctx, const clc_keypoint* h_kps, int n, uint8_t* h_desc)
{
    if (!ctx || n < 0 || (n > 0 && (!h_kps || !h_desc))) return fail(ctx, CLC_ERR_BAD_ARG, "describe: bad argument");
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "describe: context created without detector options");
    if ((uint32_t)n > ctx->dopts.maxkp) return fail(ctx, CLC_ERR_CAPACITY, "describe: more keypoints than DetectorOptions.maxkp");
    if (!ctx->pyramid_valid) return fail(ctx, CLC_ERR_STATE, "describe before pyramid_build");
    if (n == 0) return CLC_OK;
    for (int i = 0; i < n; ++i)
        if (h_kps[i].scale >= ctx->pd.levels) return fail(ctx, CLC_ERR_BAD_ARG, "describe: keypoint scale >= scale_levels");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    CLC_HIP(ctx, hipMemcpyAsync(ctx->d_kps, h_kps, (size_t)n * sizeof(clc_keypoint), hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, launch_clatch(ctx->pd, ctx->d_arena, ctx->d_kps, n, ctx->d_desc, ctx->stream, &ctx->prof));
    CLC_HIP(ctx, hipMemcpyAsync(h_desc, ctx->d_desc, (size_t)n * CLC_DESC_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

int clc_keypoints_to_features(const clc_keypoint* h_kps, int n, float* h_feat4)
{
    if (n < 0 || (n > 0 && (!h_kps || !h_feat4))) return CLC_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i) {
        // GPUDetector.hpp:173: static_cast<float>(std::pow(1.2f, kps[i].scale)) -- pow(float, integer) is evaluated in double
        const float scale = (float)std::pow((double)1.2f, (double)h_kps[i].scale);
        h_feat4[4 * i + 0] = scale * (float)h_kps[i].x;
        h_feat4[4 * i + 1] = scale * (float)h_kps[i].y;
        h_feat4[4 * i + 2] = 7.0f * scale;
        h_feat4[4 * i + 3] = h_kps[i].angle;
    }
    return CLC_OK;
}

/* ---- match ---------------------------------------------------------------------------------- */

int clc_k2nn_set_formulation(clc_ctx* ctx, int formulation)
{
    if (!ctx || (formulation != CLC_K2NN_MATRIX && formulation != CLC_K2NN_POPCOUNT))
        return fail(ctx, CLC_ERR_BAD_ARG, "k2nn_set_formulation: bad argument");
    ctx->formulation = formulation == CLC_K2NN_POPCOUNT ? K2NN_POPCOUNT : K2NN_MATRIX;
    return CLC_OK;
}

int clc_k2nn_queries_per_block(const clc_ctx* ctx)
{
    return k2nn_queries_per_block(ctx ? ctx->formulation : K2NN_MATRIX);
}

int clc_k2nn_plan_query(const clc_ctx* ctx, int nq, int nt, int32_t* info)
{
    if (!ctx || nq < 0 || nt < 0 || !info) return CLC_ERR_BAD_ARG;
    K2nnJobDev jb{};
    jb.nq = (uint32_t)nq; jb.nt = (uint32_t)nt;
    const int target = ctx->target_blocks > 0 ? ctx->target_blocks : default_target_blocks(ctx->formulation, ctx->k2dev);
    const K2nnPlan plan = k2nn_plan(&jb, 1, target, ctx->xcd_map, ctx->formulation, ctx->bias_a, ctx->bias_b, ctx->k2dev);
    info[0] = (int32_t)jb.qblocks; info[1] = (int32_t)jb.splits; info[2] = (int32_t)jb.t_per_split; info[3] = plan.atomic_merge ? 1 : 0;
    info[4] = (int32_t)jb.bias_a; info[5] = (int32_t)jb.bias_b; info[6] = k2nn_queries_per_block(ctx->formulation); info[7] = target;
    return CLC_OK;
}

int clc_k2nn_device_info(const clc_ctx* ctx, int32_t* info, float* probe_us)
{
    if (!ctx || !info) return CLC_ERR_BAD_ARG;
    info[0] = (int32_t)ctx->k2dev.n_xcd; info[1] = (int32_t)ctx->k2dev.n_cu; info[2] = default_target_blocks(ctx->formulation, ctx->k2dev);
    info[3] = ctx->bias_a; info[4] = ctx->bias_b; info[5] = ctx->bias_source; info[6] = (int32_t)kK2nnXcds; info[7] = ctx->target_blocks;
    if (probe_us) for (int c = 0; c < 4; ++c) probe_us[c] = ctx->bias_probe_us[c];
    return CLC_OK;
}

int clc_k2nn_clock_check(clc_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, int32_t* d_match, void* stream,
                         double* ghz_median, double* ghz_min, double* ghz_max, int* workgroups)
{
    if (!ctx || nq <= 0 || nt <= 0 || !d_q || !d_t || !d_match) return fail(ctx, CLC_ERR_BAD_ARG, "k2nn_clock_check: bad argument");
    if (((uintptr_t)d_q & 15u) || ((uintptr_t)d_t & 15u)) return fail(ctx, CLC_ERR_BAD_ARG, "k2nn_clock_check: misaligned device pointer");
    if (ctx->formulation == K2NN_POPCOUNT) return fail(ctx, CLC_ERR_STATE, "k2nn_clock_check: matrix formulation only");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    std::vector<K2nnJobDev> jobs(1);
    jobs[0] = K2nnJobDev{};
    jobs[0].q = (const uint4*)d_q; jobs[0].t = (const uint4*)d_t; jobs[0].out = d_match;
    jobs[0].nq = (uint32_t)nq; jobs[0].nt = (uint32_t)nt; jobs[0].thr = 40u;
    const int target = ctx->target_blocks > 0 ? ctx->target_blocks : default_target_blocks(ctx->formulation, ctx->k2dev);
    const K2nnPlan plan = k2nn_plan(jobs.data(), 1, target, ctx->xcd_map, ctx->formulation, ctx->bias_a, ctx->bias_b, ctx->k2dev);
    if (!plan.atomic_merge) return fail(ctx, CLC_ERR_CAPACITY, "k2nn_clock_check: train set too large");
    int rc = ensure_partial(ctx, plan.partial_elems);
    if (rc != CLC_OK) return rc;
    if (ctx->partial_dirty) {
        CLC_HIP(ctx, hipMemsetAsync(ctx->d_partial, 0xFF, ctx->partial_cap * sizeof(uint2), st));
        ctx->partial_dirty = false;
    }
    // the launch grid pads the query blocks to a multiple of 8 (XCD-dealt order, k2nn.hip) and a stamped workgroup
    // writes row blockIdx.y * gridDim.x + blockIdx.x: size the buffer for the PADDED grid (rows of padding workgroups
    // stay zero and are skipped below)
    const size_t nwg = (size_t)((jobs[0].qblocks + 7u) & ~7u) * jobs[0].splits;
    uint64_t* d_stamps = nullptr;
    CLC_HIP(ctx, hipMalloc((void**)&d_stamps, nwg * 8 * sizeof(uint64_t)));
    std::vector<uint64_t> h(nwg * 8);
    hipError_t e = hipMemsetAsync(d_stamps, 0, nwg * 8 * sizeof(uint64_t), st);
    if (e == hipSuccess) e = launch_k2nn(jobs.data(), 1, ctx->d_partial, st, nullptr, ctx->formulation, d_stamps);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_stamps, nwg * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_stamps);
    if (e != hipSuccess) return fail(ctx, CLC_ERR_HIP, "k2nn_clock_check", e);
    if (const char* dump = getenv("CLC_K2NN_STAMP_DUMP")) {            // diagnostic: raw per-workgroup stamps for tools/k2nn_timeline.py
        if (FILE* f = fopen(dump, "wb")) { fwrite(h.data(), sizeof(uint64_t), h.size(), f); fclose(f); }
    }
    std::vector<double> ghz;
    for (size_t w = 0; w < nwg; ++w) {
        const uint64_t dc = h[8 * w + 2] - h[8 * w], dr = h[8 * w + 3] - h[8 * w + 1];
        if (dr > 20) ghz.push_back((double)dc / (double)dr * 0.1);       // s_memrealtime ticks at 100 MHz
    }
    if (ghz.empty()) return fail(ctx, CLC_ERR_STATE, "k2nn_clock_check: sweep too short to stamp");
    std::sort(ghz.begin(), ghz.end());
    if (ghz_median) *ghz_median = ghz[ghz.size() / 2];
    if (ghz_min) *ghz_min = ghz.front();
    if (ghz_max) *ghz_max = ghz.back();
    if (workgroups) *workgroups = (int)ghz.size();
    return CLC_OK;
}

int clc_match_2nn_dev(clc_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, int threshold,
                      int32_t* d_match, void* stream)
{
    if (!ctx || nq < 0 || nt < 0 || (nq > 0 && (!d_q || !d_match)) || (nt > 0 && !d_t))
        return fail(ctx, CLC_ERR_BAD_ARG, "match_2nn: bad argument");
    if (((uintptr_t)d_q & 15u) || ((uintptr_t)d_t & 15u) || ((uintptr_t)d_match & 3u))
        return fail(ctx, CLC_ERR_BAD_ARG, "match_2nn: device pointers must be 16-byte aligned");
    if (nq == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<K2nnJobDev> jobs(1);
    jobs[0] = K2nnJobDev{};
    jobs[0].q = (const uint4*)d_q;
    jobs[0].t = (const uint4*)d_t;
    jobs[0].out = d_match;
    jobs[0].nq = (uint32_t)nq;
    jobs[0].nt = (uint32_t)nt;
    jobs[0].thr = (uint32_t)(uint8_t)threshold;   // CUDAK2NN.cu:46: the kernel parameter is uint8_t
    return run_jobs(ctx, jobs, pick(ctx, stream));
}

/* ---- describe both cameras of a pair and match them, as ONE step ------------------------------------------------------------
 * The reference's call pattern for a pair is detectAndDescribe of each camera (GPUDetector.hpp:216-291) and then
 * computeMatchesPair (GPUMatcher.hpp:165-172, :180-226).  Here: one pyramid launch and one CLATCH launch for both cameras, one sweep
 * launch, on the caller's stream.  (Round 5 also shipped a chunked form -- sweeps over finished chunks of the query camera on a
 * second stream behind one-wave gates that polled the describe launch's progress counters; it measured slower at every chunking
 * on MI355X and is gone: profiles/r05_step_overlap.txt, profiles/r06_removed_variants.patch.) */
int clc_describe_match_pair_dev(clc_ctx* ctx, const void* const* d_imgs, uint32_t width, uint32_t height, size_t pitch,
                                const clc_keypoint* const* d_kps, const int* counts, void* const* d_desc, int threshold,
                                int32_t* d_match, void* stream)
{
    if (!ctx || !d_imgs || !d_kps || !counts || !d_desc || !d_imgs[0] || !d_imgs[1] || counts[0] < 0 || counts[1] < 0 ||
        (counts[0] > 0 && (!d_kps[0] || !d_desc[0] || !d_match)) || (counts[1] > 0 && (!d_kps[1] || !d_desc[1])))
        return fail(ctx, CLC_ERR_BAD_ARG, "describe_match_pair: bad argument");
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "describe_match_pair: context created without detector options");
    if (width != ctx->dopts.width || height != ctx->dopts.height || pitch < width || pitch > 0xFFFFFFFFull)
        return fail(ctx, CLC_ERR_BAD_ARG, "describe_match_pair: image size differs from DetectorOptions width/height");
    for (int b = 0; b < 2; ++b)
        if (((uintptr_t)d_desc[b] & 15u) || ((uintptr_t)d_kps[b] & 3u)) return fail(ctx, CLC_ERR_BAD_ARG, "describe_match_pair: misaligned device pointer");
    if ((uintptr_t)d_match & 3u) return fail(ctx, CLC_ERR_BAD_ARG, "describe_match_pair: misaligned device pointer");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    { const int rc = ensure_slots(ctx, 2, st); if (rc != CLC_OK) return rc; }
    // pyramid slot 0 = the TRAIN camera (camera 1 of the pair), slot 1 = the query camera: the describe launch dispatches slot 0 first
    const uint8_t* srcs[2] = { (const uint8_t*)d_imgs[1], (const uint8_t*)d_imgs[0] };
    ClatchBatch batch{};
    batch.kps[0] = d_kps[1]; batch.desc[0] = (uint64_t*)d_desc[1]; batch.n[0] = counts[1];
    batch.kps[1] = d_kps[0]; batch.desc[1] = (uint64_t*)d_desc[0]; batch.n[1] = counts[0];
    ctx->pyramid_valid = false;
    CLC_HIP(ctx, launch_pyramid_batch(ctx->pd, ctx->d_arena, ctx->arena_bytes, srcs, 2, (uint32_t)pitch, st, &ctx->prof));
    ctx->pyramid_valid = true;
    CLC_HIP(ctx, launch_clatch_batch(ctx->pd, ctx->d_arena, ctx->arena_bytes, batch, 2, st, &ctx->prof));
    if (counts[0] == 0) return CLC_OK;
    std::vector<K2nnJobDev> jobs(1);
    jobs[0] = K2nnJobDev{};
    jobs[0].q = (const uint4*)d_desc[0]; jobs[0].t = (const uint4*)d_desc[1]; jobs[0].out = d_match;
    jobs[0].nq = (uint32_t)counts[0]; jobs[0].nt = (uint32_t)counts[1]; jobs[0].thr = (uint32_t)(uint8_t)threshold;
    return run_jobs(ctx, jobs, st);
}

int clc_match_jobs_dev(clc_ctx* ctx, const void* d_desc_base, const clc_match_job* h_jobs, int njobs,
                       int32_t* d_match, void* stream)
{
    if (!ctx || njobs < 0 || (njobs > 0 && (!d_desc_base || !h_jobs || !d_match)))
        return fail(ctx, CLC_ERR_BAD_ARG, "match_jobs: bad argument");
    if (((uintptr_t)d_desc_base & 15u) || ((uintptr_t)d_match & 3u))
        return fail(ctx, CLC_ERR_BAD_ARG, "match_jobs: device pointers must be 16-byte aligned");
    if (njobs == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<K2nnJobDev> jobs;
    jobs.reserve(njobs);
    for (int j = 0; j < njobs; ++j) {
        if (h_jobs[j].nq == 0) continue;
        K2nnJobDev jb{};
        jb.q = (const uint4*)((const uint8_t*)d_desc_base + (size_t)h_jobs[j].q_offset * CLC_DESC_BYTES);
        jb.t = (const uint4*)((const uint8_t*)d_desc_base + (size_t)h_jobs[j].t_offset * CLC_DESC_BYTES);
        jb.out = d_match + h_jobs[j].out_offset;
        jb.nq = h_jobs[j].nq;
        jb.nt = h_jobs[j].nt;
        jb.thr = (uint32_t)(uint8_t)h_jobs[j].threshold;
        jobs.push_back(jb);
    }
    if (jobs.empty()) return CLC_OK;
    return run_jobs(ctx, jobs, pick(ctx, stream));
}

int clc_match_jobs_counted_dev(clc_ctx* ctx, const void* d_desc_base, const clc_match_job* h_jobs, int njobs,
                               const int32_t* const* d_cnt_q, const int32_t* const* d_cnt_t, const uint32_t* q_row0,
                               int32_t* d_match, void* stream)
{
    if (!ctx || njobs < 0 || (njobs > 0 && (!d_desc_base || !h_jobs || !d_match || !d_cnt_q || !d_cnt_t || !q_row0)))
        return fail(ctx, CLC_ERR_BAD_ARG, "match_jobs_counted: bad argument");
    if (((uintptr_t)d_desc_base & 15u) || ((uintptr_t)d_match & 3u))
        return fail(ctx, CLC_ERR_BAD_ARG, "match_jobs_counted: device pointers must be 16-byte aligned");
    if (njobs == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<K2nnJobDev> jobs;
    jobs.reserve(njobs);
    for (int j = 0; j < njobs; ++j) {
        if (h_jobs[j].nq == 0) continue;
        // a planned train set of zero rows can hold nothing whatever the device count says; the sweep needs at least one split to
        // answer the planned rows, so such a job is run with one (empty) planned row -- the count clamps it to zero again
        if (!d_cnt_q[j] || !d_cnt_t[j]) return fail(ctx, CLC_ERR_BAD_ARG, "match_jobs_counted: null count pointer");
        K2nnJobDev jb{};
        jb.q = (const uint4*)((const uint8_t*)d_desc_base + (size_t)h_jobs[j].q_offset * CLC_DESC_BYTES);
        jb.t = (const uint4*)((const uint8_t*)d_desc_base + (size_t)h_jobs[j].t_offset * CLC_DESC_BYTES);
        jb.out = d_match + h_jobs[j].out_offset;
        jb.nq = h_jobs[j].nq;
        jb.nt = h_jobs[j].nt ? h_jobs[j].nt : 1u;
        jb.thr = (uint32_t)(uint8_t)h_jobs[j].threshold;
        jb.cnt_q = d_cnt_q[j]; jb.cnt_t = d_cnt_t[j]; jb.q_row0 = q_row0[j];
        jobs.push_back(jb);
    }
    if (jobs.empty()) return CLC_OK;
    return run_jobs(ctx, jobs, pick(ctx, stream));
}

// ---- device-side descriptor cache ------------------------------------------------------------------------------------------
// The reference's host flow hands descriptors from the detector to the matcher through host memory (FeatureMap regions:
// GPUDetector.hpp:181 -> GPUMatcher.hpp:188-196), and GPUMatcher uploads them again for every call.  Here the detector can PUBLISH
// the device copy of a block it has just written to a host address (clc_desc_cache_publish); a host-pointer match entry point that
// is later given that address finds the rows on the device and skips the upload.  How far a lookup trusts an entry is a property of
// the context that looks it up (clc_desc_cache_mode):
//   VERIFY (default): address and count match AND a 64-bit position-keyed fold of ALL rows of the host block equals the one taken at
//           publish time -- a block rewritten anywhere, or an allocation reused for other rows, can only miss (round 4 compared the
//           first and last row only: a block edited in the middle was answered with the stale device rows);
//   TRUST:  address, count and 18 sampled rows (first, last, 16 spread over the block) -- for a host that OWNS its blocks and states
//           that it does not rewrite a published block in place (HIPDetector / HIPMatcher over FeatureMap regions, which only the
//           detector writes); no pass over the block, the lookup costs a microsecond;
//   OFF:    every block is uploaded, as the reference does.
// Process-wide, per device, mutex-protected; entries in use by a running call are never evicted.  CLC_DESC_CACHE=0|verify|trust in the
// environment sets the mode new contexts start with.
enum { kCacheSamples = 16 };
struct DescCacheEntry {
    int device = -1;
    const void* h = nullptr;
    int n = 0;
    uint8_t first[CLC_DESC_BYTES], last[CLC_DESC_BYTES];
    uint8_t sample[kCacheSamples][CLC_DESC_BYTES];      // rows (i + 1) * n / (kCacheSamples + 1)
    uint64_t fold = 0;       // desc_block_fold of the n published rows
    bool has_fold = false;   // (blocks published by a TRUST context carry none: only TRUST lookups can hit them)
    uint8_t* d = nullptr;
    size_t cap = 0;          // rows allocated
    uint64_t stamp = 0;
    int busy = 0;
};
struct DescCache {
    std::mutex mu;
    std::vector<DescCacheEntry> e;
    uint64_t clock = 0;
    uint64_t hits = 0, misses = 0;      // lookups answered from the cache / uploaded
};
static constexpr size_t kDescCacheEntries = 32;
static DescCache& desc_cache() { static DescCache c; return c; }

// 64-bit fold of n descriptor rows: every 16 bytes are keyed with their position (the key steps per row, the four pairs of a row take
// different constants), multiplied 64 x 64 -> 128 and folded; the row terms are summed.  One multiply per 16 bytes: a 640 KB block
// takes what reading it takes.  Change detection, not cryptography.
static inline uint64_t fold_mul(const uint64_t a, const uint64_t b)
{
    const unsigned __int128 p = (unsigned __int128)a * b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}
static uint64_t desc_block_fold(const void* h, const size_t n)
{
    const uint8_t* p = (const uint8_t*)h;
    uint64_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, k = 0x9E3779B97F4A7C15ull;
    for (size_t r = 0; r < n; ++r, p += CLC_DESC_BYTES, k += 0xD1B54A32D192ED03ull) {
        uint64_t w[8];
        memcpy(w, p, sizeof w);                          // (host blocks carry no alignment promise)
        h0 += fold_mul(w[0] ^ k, w[1] ^ 0x8BB84B93962EACC9ull);
        h1 += fold_mul(w[2] ^ (k + 0x2D358DCCAA6C78A5ull), w[3] ^ 0x4B33A62ED433D4A3ull);
        h2 += fold_mul(w[4] ^ (k + 0x4D5A2DA51DE1AA47ull), w[5] ^ 0xA0761D6478BD642Full);
        h3 += fold_mul(w[6] ^ (k + 0xE7037ED1A0B428DBull), w[7] ^ 0x589965CC75374CC3ull);
    }
    return (h0 ^ (h1 << 1 | h1 >> 63)) + (h2 ^ (h3 << 7 | h3 >> 57)) + (uint64_t)n;
}
static inline const uint8_t* cache_sample_row(const void* h, const int n, const int i)
{
    return (const uint8_t*)h + (size_t)((uint64_t)(i + 1) * (uint64_t)n / (kCacheSamples + 1)) * CLC_DESC_BYTES;
}

// device rows of host block (h, n) if published and, by the rule of `mode`, still the published rows; the entry is pinned until cache_release
static const uint8_t* cache_acquire(const int mode, const int device, const void* h, const int n, DescCacheEntry** held)
{
    *held = nullptr;
    DescCache& c = desc_cache();
    if (mode == CLC_DESC_CACHE_OFF || !h || n <= 0) return nullptr;
    uint64_t fold = 0;
    bool folded = false;
    std::unique_lock<std::mutex> lk(c.mu);
    for (DescCacheEntry& en : c.e) {
        if (en.device != device || en.h != h || en.n != n || !en.d) continue;
        const uint8_t* hb = (const uint8_t*)h;
        bool same = memcmp(hb, en.first, CLC_DESC_BYTES) == 0 && memcmp(hb + (size_t)(n - 1) * CLC_DESC_BYTES, en.last, CLC_DESC_BYTES) == 0;
        for (int i = 0; same && i < kCacheSamples; ++i) same = memcmp(cache_sample_row(h, n, i), en.sample[i], CLC_DESC_BYTES) == 0;
        if (same && mode == CLC_DESC_CACHE_VERIFY) {
            if (!en.has_fold) continue;                  // published without a fold: not for a verifying context
            if (!folded) {                               // the pass over the block, outside the lock (an entry cannot move; its fields
                ++en.busy;                               // are re-read under the lock below)
                lk.unlock();
                fold = desc_block_fold(h, (size_t)n);
                folded = true;
                lk.lock();
                --en.busy;
                if (en.device != device || en.h != h || en.n != n || !en.d || !en.has_fold) continue;
            }
            same = fold == en.fold;
        }
        if (!same) {
            if (en.busy == 0) en.h = nullptr;            // the host block has been rewritten: forget the entry
            continue;                                    // (a newer entry for the same address may follow)
        }
        en.stamp = ++c.clock;
        ++en.busy;
        ++c.hits;
        *held = &en;
        return en.d;
    }
    ++c.misses;
    return nullptr;
}
static void cache_release(DescCacheEntry* held)
{
    if (!held) return;
    std::lock_guard<std::mutex> lk(desc_cache().mu);
    --held->busy;
}

static int match_host_impl(clc_ctx* ctx, const uint8_t* d_query, int nq, const uint8_t* d_train, int nt, int threshold,
                           int32_t* h_match, uint16_t* h_best, uint16_t* h_second)
{
    std::vector<K2nnJobDev> jobs(1);
    jobs[0] = K2nnJobDev{};
    jobs[0].q = (const uint4*)d_query;
    jobs[0].t = (const uint4*)d_train;
    jobs[0].out = ctx->d_match;
    jobs[0].best_out = h_best ? ctx->d_best : nullptr;
    jobs[0].second_out = h_second ? ctx->d_second : nullptr;
    jobs[0].nq = (uint32_t)nq;
    jobs[0].nt = (uint32_t)nt;
    jobs[0].thr = (uint32_t)(uint8_t)threshold;
    const int rc = run_jobs(ctx, jobs, ctx->stream);
    if (rc != CLC_OK) return rc;
    CLC_HIP(ctx, hipMemcpyAsync(h_match, ctx->d_match, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (h_best) CLC_HIP(ctx, hipMemcpyAsync(h_best, ctx->d_best, (size_t)nq * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
    if (h_second) CLC_HIP(ctx, hipMemcpyAsync(h_second, ctx->d_second, (size_t)nq * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

// query rows from the cache when the detector published them, else uploaded; the cache entry stays pinned for the duration of the call
static int match_host(clc_ctx* ctx, const void* h_q, int nq, const uint8_t* d_train, int nt, int threshold,
                      int32_t* h_match, uint16_t* h_best, uint16_t* h_second)
{
    DescCacheEntry* held = nullptr;
    const uint8_t* d_query = cache_acquire(ctx->cache_mode, ctx->device, h_q, nq, &held);
    if (!d_query) {
        const hipError_t e = hipMemcpyAsync(ctx->d_q, h_q, (size_t)nq * CLC_DESC_BYTES, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(ctx, CLC_ERR_HIP, "match: query upload", e);
        d_query = ctx->d_q;
    }
    const int rc = match_host_impl(ctx, d_query, nq, d_train, nt, threshold, h_match, h_best, h_second);
    cache_release(held);
    return rc;
}

int clc_desc_cache_publish(clc_ctx* ctx, const void* d_src, const void* h_desc, int n)
{
    if (!ctx || !h_desc || n < 0) return fail(ctx, CLC_ERR_BAD_ARG, "desc_cache_publish: bad argument");
    if (!d_src) {
        if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "desc_cache_publish: context created without detector options and no device source given");
        d_src = ctx->d_desc;
    }
    DescCache& c = desc_cache();
    if (ctx->cache_mode == CLC_DESC_CACHE_OFF || n == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    // a verifying context folds the block it publishes (the host has just written it: the pass runs out of its caches); a trusting
    // one does not, and its entries can then only be hit by trusting lookups
    const bool with_fold = ctx->cache_mode == CLC_DESC_CACHE_VERIFY;
    const uint64_t fold = with_fold ? desc_block_fold(h_desc, (size_t)n) : 0u;
    DescCacheEntry* slot = nullptr;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        for (DescCacheEntry& en : c.e)
            if (en.device == ctx->device && en.h == h_desc && en.busy == 0) { slot = &en; break; }      // the same host block again
        if (!slot && c.e.size() < kDescCacheEntries) {
            c.e.reserve(kDescCacheEntries);                                                             // entries never move (held pointers)
            c.e.emplace_back();
            slot = &c.e.back();
        }
        if (!slot) {
            for (DescCacheEntry& en : c.e)
                if (en.busy == 0 && (!slot || en.stamp < slot->stamp)) slot = &en;                        // least recently used, not in use
        }
        if (!slot) return CLC_OK;                                                                        // everything in use: do not cache
        slot->busy = 1;                                                                                  // reserved while it is being filled
        slot->h = nullptr;
    }
    hipError_t e = hipSuccess;
    if (slot->cap < (size_t)n || slot->device != ctx->device) {
        if (slot->d) (void)hipFree(slot->d);
        slot->d = nullptr; slot->cap = 0;
        e = hipMalloc((void**)&slot->d, (size_t)n * CLC_DESC_BYTES);
        if (e == hipSuccess) slot->cap = (size_t)n;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(slot->d, d_src, (size_t)n * CLC_DESC_BYTES, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    {
        std::lock_guard<std::mutex> lk(c.mu);
        slot->busy = 0;
        if (e == hipSuccess) {
            slot->device = ctx->device; slot->h = h_desc; slot->n = n; slot->stamp = ++c.clock;
            memcpy(slot->first, h_desc, CLC_DESC_BYTES);
            memcpy(slot->last, (const uint8_t*)h_desc + (size_t)(n - 1) * CLC_DESC_BYTES, CLC_DESC_BYTES);
            for (int i = 0; i < kCacheSamples; ++i) memcpy(slot->sample[i], cache_sample_row(h_desc, n, i), CLC_DESC_BYTES);
            slot->fold = fold; slot->has_fold = with_fold;
        }
    }
    if (e != hipSuccess) return fail(ctx, CLC_ERR_HIP, "desc_cache_publish", e);
    return CLC_OK;
}

int clc_desc_cache_mode(clc_ctx* ctx, int mode)
{
    if (!ctx || (mode != CLC_DESC_CACHE_OFF && mode != CLC_DESC_CACHE_VERIFY && mode != CLC_DESC_CACHE_TRUST))
        return fail(ctx, CLC_ERR_BAD_ARG, "desc_cache_mode: unknown mode");
    ctx->cache_mode = mode;
    return CLC_OK;
}

int clc_desc_cache_stats(unsigned long long* hits, unsigned long long* misses)
{
    DescCache& c = desc_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    if (hits) *hits = c.hits;
    if (misses) *misses = c.misses;
    return CLC_OK;
}

int clc_desc_cache_clear(void)
{
    DescCache& c = desc_cache();
    std::lock_guard<std::mutex> lk(c.mu);
    for (DescCacheEntry& en : c.e) {
        if (en.busy) continue;
        if (en.d) { (void)hipSetDevice(en.device); (void)hipFree(en.d); }
        en.d = nullptr; en.cap = 0; en.h = nullptr; en.n = 0;
    }
    return CLC_OK;
}

int clc_match_2nn(clc_ctx* ctx, const void* h_q, int nq, const void* h_t, int nt, int threshold,
                  int32_t* h_match, uint16_t* h_best, uint16_t* h_second)
{
    if (!ctx || nq < 0 || nt < 0 || (nq > 0 && (!h_q || !h_match)) || (nt > 0 && !h_t))
        return fail(ctx, CLC_ERR_BAD_ARG, "match_2nn: bad argument");
    if (!ctx->has_mat) return fail(ctx, CLC_ERR_STATE, "match_2nn: context created without matcher options");
    if ((uint32_t)nq > ctx->mopts.maxkp || (uint32_t)nt > ctx->mopts.maxkp)
        return fail(ctx, CLC_ERR_CAPACITY, "match_2nn: more descriptors than MatcherOptions.maxkp");
    if (nq == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    DescCacheEntry* held = nullptr;
    const uint8_t* d_train = nt > 0 ? cache_acquire(ctx->cache_mode, ctx->device, h_t, nt, &held) : nullptr;
    if (!d_train) {
        if (nt > 0) {
            const hipError_t e = hipMemcpyAsync(ctx->d_t, h_t, (size_t)nt * CLC_DESC_BYTES, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) return fail(ctx, CLC_ERR_HIP, "match_2nn: train upload", e);
        }
        d_train = ctx->d_t;
    }
    const int rc = match_host(ctx, h_q, nq, d_train, nt, threshold, h_match, h_best, h_second);
    cache_release(held);
    return rc;
}

int clc_match_pairs(clc_ctx* ctx, const void* const* h_desc, const int* counts, int ncams, const int* pairs,
                    int npairs, int threshold, int32_t* const* h_match)
{
    if (!ctx || ncams < 0 || npairs < 0 || (ncams > 0 && (!h_desc || !counts)) || (npairs > 0 && (!pairs || !h_match)))
        return fail(ctx, CLC_ERR_BAD_ARG, "match_pairs: bad argument");
    if (npairs == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<size_t> cam_off(ncams + 1, 0);
    for (int c = 0; c < ncams; ++c) {
        if (counts[c] < 0 || (counts[c] > 0 && !h_desc[c])) return fail(ctx, CLC_ERR_BAD_ARG, "match_pairs: bad camera entry");
        cam_off[c + 1] = cam_off[c] + (size_t)counts[c];
    }
    size_t out_rows = 0;
    for (int p = 0; p < npairs; ++p) {
        const int a = pairs[2 * p], b = pairs[2 * p + 1];
        if (a < 0 || a >= ncams || b < 0 || b >= ncams || (counts[a] > 0 && !h_match[p]))
            return fail(ctx, CLC_ERR_BAD_ARG, "match_pairs: bad pair entry");
        out_rows += (size_t)counts[a];
    }
    const size_t desc_bytes = cam_off[ncams] * CLC_DESC_BYTES;
    const size_t need = desc_bytes + out_rows * sizeof(int32_t) + 256;
    if (need > ctx->pairs_cap) {
        CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_pairs) CLC_HIP(ctx, hipFree(ctx->d_pairs));
        ctx->d_pairs = nullptr; ctx->pairs_cap = 0;
        CLC_HIP(ctx, hipMalloc((void**)&ctx->d_pairs, need));
        ctx->pairs_cap = need;
    }
    // a camera whose block the detector published on this device is read where it lies; the others are uploaded
    std::vector<const uint8_t*> cam_dev((size_t)ncams, nullptr);
    std::vector<DescCacheEntry*> held((size_t)ncams, nullptr);
    struct Release { std::vector<DescCacheEntry*>& h; ~Release() { for (DescCacheEntry* e : h) cache_release(e); } } release{ held };
    for (int c = 0; c < ncams; ++c) {
        if (counts[c] <= 0) continue;
        cam_dev[(size_t)c] = cache_acquire(ctx->cache_mode, ctx->device, h_desc[c], counts[c], &held[(size_t)c]);
        if (!cam_dev[(size_t)c]) {
            CLC_HIP(ctx, hipMemcpyAsync(ctx->d_pairs + cam_off[c] * CLC_DESC_BYTES, h_desc[c], (size_t)counts[c] * CLC_DESC_BYTES,
                                        hipMemcpyHostToDevice, ctx->stream));
            cam_dev[(size_t)c] = ctx->d_pairs + cam_off[c] * CLC_DESC_BYTES;
        }
    }
    int32_t* d_out = (int32_t*)(ctx->d_pairs + ((desc_bytes + 255) & ~(size_t)255));
    std::vector<K2nnJobDev> jobs;
    std::vector<size_t> out_off(npairs, 0);
    size_t o = 0;
    for (int p = 0; p < npairs; ++p) {
        const int a = pairs[2 * p], b = pairs[2 * p + 1];
        out_off[p] = o;
        if (counts[a] == 0) continue;
        K2nnJobDev jb{};
        jb.q = (const uint4*)cam_dev[(size_t)a];
        jb.t = (const uint4*)(counts[b] > 0 ? cam_dev[(size_t)b] : ctx->d_pairs);
        jb.out = d_out + o;
        jb.nq = (uint32_t)counts[a];
        jb.nt = (uint32_t)counts[b];
        jb.thr = (uint32_t)(uint8_t)threshold;
        jobs.push_back(jb);
        o += (size_t)counts[a];
    }
    if (!jobs.empty()) {
        const int rc = run_jobs(ctx, jobs, ctx->stream);
        if (rc != CLC_OK) return rc;
    }
    for (int p = 0; p < npairs; ++p) {
        const int a = pairs[2 * p];
        if (counts[a] > 0)
            CLC_HIP(ctx, hipMemcpyAsync(h_match[p], d_out + out_off[p], (size_t)counts[a] * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

int clc_set_map(clc_ctx* ctx, const void* h_desc, int n)
{
    if (!ctx || n < 0 || (n > 0 && !h_desc)) return fail(ctx, CLC_ERR_BAD_ARG, "set_map: bad argument");
    if (!ctx->has_mat) return fail(ctx, CLC_ERR_STATE, "set_map: context created without matcher options");
    if ((uint32_t)n > ctx->mopts.maxkp) return fail(ctx, CLC_ERR_CAPACITY, "set_map: more descriptors than MatcherOptions.maxkp");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    if (n > 0) CLC_HIP(ctx, hipMemcpyAsync(ctx->d_m, h_desc, (size_t)n * CLC_DESC_BYTES, hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->map_n = n;
    return CLC_OK;
}

int clc_match_map(clc_ctx* ctx, const void* h_q, int nq, int threshold, int32_t* h_match)
{
    if (!ctx || nq < 0 || (nq > 0 && (!h_q || !h_match))) return fail(ctx, CLC_ERR_BAD_ARG, "match_map: bad argument");
    if (!ctx->has_mat) return fail(ctx, CLC_ERR_STATE, "match_map: context created without matcher options");
    if (ctx->map_n < 0) return fail(ctx, CLC_ERR_STATE, "match_map before set_map");
    if ((uint32_t)nq > ctx->mopts.maxkp) return fail(ctx, CLC_ERR_CAPACITY, "match_map: more descriptors than MatcherOptions.maxkp");
    if (nq == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    return match_host(ctx, h_q, nq, ctx->d_m, ctx->map_n, threshold, h_match, nullptr, nullptr);
}

int clc_match_map_dev(clc_ctx* ctx, const void* d_q, int nq, int threshold, int32_t* d_match, void* stream)
{
    if (!ctx || nq < 0 || (nq > 0 && (!d_q || !d_match))) return fail(ctx, CLC_ERR_BAD_ARG, "match_map_dev: bad argument");
    if (!ctx->has_mat) return fail(ctx, CLC_ERR_STATE, "match_map_dev: context created without matcher options");
    if (ctx->map_n < 0) return fail(ctx, CLC_ERR_STATE, "match_map_dev before set_map");
    return clc_match_2nn_dev(ctx, d_q, nq, ctx->d_m, ctx->map_n, threshold, d_match, stream);
}

/* ---- PnP ------------------------------------------------------------------------------------ */

static int pnp_upload(clc_ctx* ctx, const double* h_Rt, int H, const double* h_X, const double* h_x, int N,
                      const double* h_K, size_t extra, double** dRt, double** dX, double** dx, double** dK, double** dExtra)
{
    const size_t need = (size_t)12 * H + (size_t)5 * N + 16 + extra;
    const int rc = ensure_pnp(ctx, need);
    if (rc != CLC_OK) return rc;
    double* p = ctx->d_pnp;
    *dRt = p; p += (size_t)12 * H;
    *dX = p; p += (size_t)3 * N;
    *dx = p; p += (size_t)2 * N;
    *dK = p; p += 16;
    *dExtra = p;
    CLC_HIP(ctx, hipMemcpyAsync(*dRt, h_Rt, sizeof(double) * 12 * H, hipMemcpyHostToDevice, ctx->stream));
    if (N > 0) {
        CLC_HIP(ctx, hipMemcpyAsync(*dX, h_X, sizeof(double) * 3 * N, hipMemcpyHostToDevice, ctx->stream));
        CLC_HIP(ctx, hipMemcpyAsync(*dx, h_x, sizeof(double) * 2 * N, hipMemcpyHostToDevice, ctx->stream));
    }
    CLC_HIP(ctx, hipMemcpyAsync(*dK, h_K, sizeof(double) * 9, hipMemcpyHostToDevice, ctx->stream));
    return CLC_OK;
}

int clc_pnp_residuals(clc_ctx* ctx, const double* h_Rt, int H, const double* h_X, const double* h_x, int N,
                      const double* h_K, double* h_err)
{
    if (!ctx || H < 0 || N < 0 || !h_K || (H > 0 && !h_Rt) || (N > 0 && (!h_X || !h_x)) || (H > 0 && N > 0 && !h_err))
        return fail(ctx, CLC_ERR_BAD_ARG, "pnp_residuals: bad argument");
    if (H == 0 || N == 0) return CLC_OK;
    if (H > 65535) return fail(ctx, CLC_ERR_CAPACITY, "pnp_residuals: more than 65535 hypotheses per call");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    double *dRt, *dX, *dx, *dK, *dE;
    const int rc = pnp_upload(ctx, h_Rt, H, h_X, h_x, N, h_K, (size_t)H * N, &dRt, &dX, &dx, &dK, &dE);
    if (rc != CLC_OK) return rc;
    CLC_HIP(ctx, launch_pnp_residuals(dRt, H, dX, dx, N, dK, dE, ctx->stream, &ctx->prof));
    CLC_HIP(ctx, hipMemcpyAsync(h_err, dE, sizeof(double) * (size_t)H * N, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

int clc_pnp_score(clc_ctx* ctx, const double* h_Rt, int H, const double* h_X, const double* h_x, int N,
                  const double* h_K, double thr2, int32_t* h_count, double* h_cost)
{
    if (!ctx || H < 0 || N < 0 || !h_K || (H > 0 && !h_Rt) || (N > 0 && (!h_X || !h_x)))
        return fail(ctx, CLC_ERR_BAD_ARG, "pnp_score: bad argument");
    if (H == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    double *dRt, *dX, *dx, *dK, *dE;
    const int rc = pnp_upload(ctx, h_Rt, H, h_X, h_x, N, h_K, (size_t)2 * H, &dRt, &dX, &dx, &dK, &dE);
    if (rc != CLC_OK) return rc;
    double* d_cost = dE;
    int32_t* d_count = (int32_t*)(dE + H);
    CLC_HIP(ctx, launch_pnp_score(dRt, H, dX, dx, N, dK, thr2, d_count, d_cost, ctx->stream, &ctx->prof));
    if (h_cost) CLC_HIP(ctx, hipMemcpyAsync(h_cost, d_cost, sizeof(double) * H, hipMemcpyDeviceToHost, ctx->stream));
    if (h_count) CLC_HIP(ctx, hipMemcpyAsync(h_count, d_count, sizeof(int32_t) * H, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

static int epipolar_impl(clc_ctx* ctx, const double* h_F, int H, const double* h_x1, const double* h_x2, int N, double thr2,
                         double* h_err, int32_t* h_count, double* h_cost)
{
    if (!ctx || H < 0 || N < 0 || (H > 0 && !h_F) || (N > 0 && (!h_x1 || !h_x2))) return fail(ctx, CLC_ERR_BAD_ARG, "epipolar: bad argument");
    if (H == 0 || (N == 0 && h_err)) return CLC_OK;
    if (H > 65535) return fail(ctx, CLC_ERR_CAPACITY, "epipolar: more than 65535 hypotheses per call");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t out = h_err ? (size_t)H * N : (size_t)2 * H;
    const int rc = ensure_pnp(ctx, (size_t)9 * H + (size_t)4 * N + out + 8);
    if (rc != CLC_OK) return rc;
    double* dF = ctx->d_pnp;
    double* d1 = dF + (size_t)9 * H;
    double* d2 = d1 + (size_t)2 * N;
    double* dO = d2 + (size_t)2 * N;
    CLC_HIP(ctx, hipMemcpyAsync(dF, h_F, sizeof(double) * 9 * H, hipMemcpyHostToDevice, ctx->stream));
    if (N > 0) {
        CLC_HIP(ctx, hipMemcpyAsync(d1, h_x1, sizeof(double) * 2 * N, hipMemcpyHostToDevice, ctx->stream));
        CLC_HIP(ctx, hipMemcpyAsync(d2, h_x2, sizeof(double) * 2 * N, hipMemcpyHostToDevice, ctx->stream));
    }
    if (h_err) {
        CLC_HIP(ctx, launch_epipolar(dF, H, d1, d2, N, thr2, dO, nullptr, nullptr, ctx->stream, &ctx->prof));
        CLC_HIP(ctx, hipMemcpyAsync(h_err, dO, sizeof(double) * (size_t)H * N, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        int32_t* dC = (int32_t*)(dO + H);
        CLC_HIP(ctx, launch_epipolar(dF, H, d1, d2, N, thr2, nullptr, dC, dO, ctx->stream, &ctx->prof));
        if (h_cost) CLC_HIP(ctx, hipMemcpyAsync(h_cost, dO, sizeof(double) * H, hipMemcpyDeviceToHost, ctx->stream));
        if (h_count) CLC_HIP(ctx, hipMemcpyAsync(h_count, dC, sizeof(int32_t) * H, hipMemcpyDeviceToHost, ctx->stream));
    }
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CLC_OK;
}

// S minimal samples of `k` distinct indices in [0, N) from a xorshift64* stream
static void draw_samples(uint64_t seed, int S, int N, std::vector<int32_t>& out, int k = 3)
{
    out.resize((size_t)k * S);
    uint64_t st = seed ? seed : 0x9E3779B97F4A7C15ull;
    auto next = [&]() { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st * 0x2545F4914F6CDD1Dull; };   // xorshift64*
    for (int s = 0; s < S; ++s) {
        for (int j = 0; j < k; ++j) {
            int32_t v;
            bool again;
            do {
                v = (int32_t)(next() % (uint64_t)N);
                again = false;
                for (int m = 0; m < j && N > j; ++m) again = again || out[(size_t)k * s + m] == v;
            } while (again);
            out[(size_t)k * s + j] = v;
        }
    }
}

// Host staging for the robust pose solve: ONE pinned buffer, ONE H2D copy in, ONE D2H copy out.
static int ensure_pinned(clc_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->pin_cap) return CLC_OK;
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_pin) CLC_HIP(ctx, hipHostFree(ctx->h_pin));
    ctx->h_pin = nullptr; ctx->pin_cap = 0;
    const size_t cap = bytes + bytes / 2;           // (see ensure_pnp: a pinned allocation costs a millisecond)
    CLC_HIP(ctx, hipHostMalloc(&ctx->h_pin, cap, hipHostMallocDefault));
    ctx->pin_cap = cap;
    return CLC_OK;
}

static int pnp_ransac_impl(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K,
                           const int32_t* h_samples, int S, uint64_t seed, double thr2, double* h_Rt, uint8_t* h_mask,
                           int* n_inliers, double* cost, double* h_all_Rt, double refine_huber = -1.0, double* h_cov = nullptr,
                           double* rmse = nullptr)
{
    if (!ctx || N < 0 || S < 0 || !h_K || (N > 0 && (!h_X || !h_x))) return fail(ctx, CLC_ERR_BAD_ARG, "pnp_ransac: bad argument");
    if (n_inliers) *n_inliers = 0;
    if (N < 3 || S == 0) { if (h_mask && N > 0) memset(h_mask, 0, (size_t)N); return CLC_OK; }
    if (S > 16384) return fail(ctx, CLC_ERR_CAPACITY, "pnp_ransac: more than 16384 samples per call");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> drawn;
    if (!h_samples) { draw_samples(seed, S, N, drawn); h_samples = drawn.data(); }
    // device workspace, in doubles:  [ X 3N | x 2N | K 16 | samples (3S int32) ]  <- one H2D copy
    //                               [ Rt 48S | cost 4S | count (4S int32) ]      scratch
    //                               [ result record | mask N bytes ]             <- one D2H copy
    const size_t in_d = (size_t)5 * N + 16 + ((size_t)3 * S + 1) / 2;
    const size_t scr_d = (size_t)48 * S + (size_t)4 * S + ((size_t)4 * S + 1) / 2;
    const bool refine = refine_huber > 0.0;
    const size_t ref_d = refine ? (pnp_refine_out_bytes() + 7) / 8 : 0;      // refine record rides in front of the mask
    const size_t res_d = (pnp_result_bytes() + 7) / 8 + ref_d;
    const size_t out_d = res_d + ((size_t)N + 7) / 8;
    int rc = ensure_pnp(ctx, in_d + scr_d + out_d + 8);
    if (rc != CLC_OK) return rc;
    rc = ensure_pinned(ctx, (in_d + out_d) * sizeof(double) + 64);
    if (rc != CLC_OK) return rc;
    double* dX = ctx->d_pnp;
    double* dx = dX + (size_t)3 * N;
    double* dK = dx + (size_t)2 * N;
    int32_t* dSamples = (int32_t*)(dK + 16);
    double* dRt = ctx->d_pnp + in_d;
    double* dCost = dRt + (size_t)48 * S;
    int32_t* dCount = (int32_t*)(dCost + (size_t)4 * S);
    double* dRes = ctx->d_pnp + in_d + scr_d;
    uint8_t* dMask = (uint8_t*)(dRes + res_d);
    // Inputs go into the pinned buffer and stay there: the first launch reads them over PCIe and stages them into
    // device memory itself, the last launch writes record + mask back into the pinned buffer.  No copy commands: a
    // pose solve is three (four with refinement) kernel launches and one stream synchronisation.
    double* hp = (double*)ctx->h_pin;
    double* hout = hp + in_d;
    memcpy(hp, h_X, sizeof(double) * 3 * N);
    memcpy(hp + (size_t)3 * N, h_x, sizeof(double) * 2 * N);
    memcpy(hp + (size_t)5 * N, h_K, sizeof(double) * 9);
    memcpy(hp + (size_t)5 * N + 16, h_samples, sizeof(int32_t) * 3 * S);
    const size_t ref_off = (pnp_result_bytes() + 7) / 8;
    PnpHostStage hs;
    hs.src = hp;
    hs.n_doubles = (int)in_d;
    hs.h_mask = (uint8_t*)(hout + res_d);
    hs.h_result = hout;
    CLC_HIP(ctx, launch_pnp_ransac(dX, dx, N, dK, dSamples, S, thr2, dRt, dCount, dCost, dMask, dRes, ctx->stream, &ctx->prof, &hs));
    if (refine)
        CLC_HIP(ctx, launch_pnp_refine(dRes /* PnpResult.Rt */, dX, dx, dMask, N, dK, refine_huber, 50, dRes + ref_off, ctx->stream,
                                       &ctx->prof, (const int32_t*)((const uint8_t*)dRes + pnp_result_valid_offset()), hout + ref_off));
    if (h_all_Rt) CLC_HIP(ctx, hipMemcpyAsync(h_all_Rt, dRt, sizeof(double) * 48 * S, hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    struct { double Rt[12]; double cost; int32_t h; int32_t count; } r;
    memcpy(&r, hout, sizeof r);
    if (h_Rt) memcpy(h_Rt, r.Rt, sizeof(double) * 12);
    if (h_mask) memcpy(h_mask, (const uint8_t*)(hout + res_d), (size_t)N);
    if (n_inliers) *n_inliers = r.h >= 0 ? r.count : 0;
    if (cost) *cost = r.cost;
    if (refine) {
        struct { double Rt[12]; double cov[36]; double cost; double rmse; int32_t iterations; int32_t n_used; } f;
        memcpy(&f, hout + ref_off, sizeof f);
        if (h_Rt) memcpy(h_Rt, f.Rt, sizeof f.Rt);
        if (h_cov) memcpy(h_cov, f.cov, sizeof f.cov);
        if (rmse) *rmse = f.rmse;
    }
    return CLC_OK;
}

int clc_epipolar_residuals(clc_ctx* ctx, const double* h_F, int H, const double* h_x1, const double* h_x2, int N, double* h_err)
{
    if (H > 0 && N > 0 && !h_err) return fail(ctx, CLC_ERR_BAD_ARG, "epipolar_residuals: null output");
    return epipolar_impl(ctx, h_F, H, h_x1, h_x2, N, 0.0, h_err, nullptr, nullptr);
}

int clc_epipolar_score(clc_ctx* ctx, const double* h_F, int H, const double* h_x1, const double* h_x2, int N, double thr2,
                       int32_t* h_count, double* h_cost)
{
    return epipolar_impl(ctx, h_F, H, h_x1, h_x2, N, thr2, nullptr, h_count, h_cost);
}

static int essential_impl(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1, const double* h_K2,
                          const int32_t* h_samples, int S, uint64_t seed, double thr2, double* h_E, double* h_F, uint8_t* h_mask,
                          int* n_inliers, double* h_all_E)
{
    if (!ctx || N < 0 || S < 0 || !h_K1 || !h_K2 || (N > 0 && (!h_x1 || !h_x2))) return fail(ctx, CLC_ERR_BAD_ARG, "essential_ransac: bad argument");
    if (n_inliers) *n_inliers = 0;
    if (N < 5 || S == 0) { if (h_mask && N > 0) memset(h_mask, 0, (size_t)N); return CLC_OK; }
    if (S > 6000) return fail(ctx, CLC_ERR_CAPACITY, "essential_ransac: more than 6000 samples per call");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> drawn;
    if (!h_samples) { draw_samples(seed, S, N, drawn, 5); h_samples = drawn.data(); }
    // device workspace in doubles: [ x1 2N | x2 2N | K1 16 | K2 16 | samples (5S int32) ] in,
    //                              [ FE 180 S | cost 10 S | count (10 S int32) ] scratch, [ result | mask ] out
    const size_t in_d = (size_t)4 * N + 32 + ((size_t)5 * S + 1) / 2;
    const size_t scr_d = (size_t)180 * S + (size_t)10 * S + ((size_t)10 * S + 1) / 2;
    const size_t res_d = (epi_result_bytes() + 7) / 8;
    const size_t out_d = res_d + ((size_t)N + 7) / 8;
    int rc = ensure_pnp(ctx, in_d + scr_d + out_d + 8);
    if (rc != CLC_OK) return rc;
    rc = ensure_pinned(ctx, (in_d > out_d ? in_d : out_d) * sizeof(double) + 64);
    if (rc != CLC_OK) return rc;
    double* d1 = ctx->d_pnp;
    double* d2 = d1 + (size_t)2 * N;
    double* dK1 = d2 + (size_t)2 * N;
    double* dK2 = dK1 + 16;
    int32_t* dSamples = (int32_t*)(dK2 + 16);
    double* dFE = ctx->d_pnp + in_d;
    double* dCost = dFE + (size_t)180 * S;
    int32_t* dCount = (int32_t*)(dCost + (size_t)10 * S);
    double* dRes = ctx->d_pnp + in_d + scr_d;
    uint8_t* dMask = (uint8_t*)(dRes + res_d);
    double* hp = (double*)ctx->h_pin;
    memcpy(hp, h_x1, sizeof(double) * 2 * N);
    memcpy(hp + (size_t)2 * N, h_x2, sizeof(double) * 2 * N);
    memcpy(hp + (size_t)4 * N, h_K1, sizeof(double) * 9);
    memcpy(hp + (size_t)4 * N + 16, h_K2, sizeof(double) * 9);
    memcpy(hp + (size_t)4 * N + 32, h_samples, sizeof(int32_t) * 5 * S);
    CLC_HIP(ctx, hipMemcpyAsync(d1, hp, in_d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, launch_essential_ransac(d1, d2, N, dK1, dK2, dSamples, S, thr2, dFE, dCount, dCost, dMask, dRes, ctx->stream, &ctx->prof));
    CLC_HIP(ctx, hipMemcpyAsync(hp, dRes, out_d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<double> fe;
    if (h_all_E) { fe.resize((size_t)180 * S); CLC_HIP(ctx, hipMemcpyAsync(fe.data(), dFE, sizeof(double) * 180 * S, hipMemcpyDeviceToHost, ctx->stream)); }
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    struct { double E[9]; double F[9]; double cost; int32_t h; int32_t count; } r;
    memcpy(&r, hp, sizeof r);
    if (h_E) memcpy(h_E, r.E, sizeof r.E);
    if (h_F) memcpy(h_F, r.F, sizeof r.F);
    if (h_mask) memcpy(h_mask, (const uint8_t*)(hp + res_d), (size_t)N);
    if (n_inliers) *n_inliers = r.h >= 0 ? r.count : 0;
    if (h_all_E)
        for (size_t k = 0; k < (size_t)10 * S; ++k) memcpy(h_all_E + 9 * k, fe.data() + 18 * k + 9, sizeof(double) * 9);
    return CLC_OK;
}

int clc_essential_ransac(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1, const double* h_K2,
                         const int32_t* h_samples, int S, uint64_t seed, double thr2, double* h_E, double* h_F,
                         uint8_t* h_inlier_mask, int* n_inliers)
{
    return essential_impl(ctx, h_x1, h_x2, N, h_K1, h_K2, h_samples, S, seed, thr2, h_E, h_F, h_inlier_mask, n_inliers, nullptr);
}

int clc_essential_fivepoint(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1, const double* h_K2,
                            const int32_t* h_samples, int S, double* h_E_out)
{
    if (!h_samples || !h_E_out) return fail(ctx, CLC_ERR_BAD_ARG, "essential_fivepoint: bad argument");
    return essential_impl(ctx, h_x1, h_x2, N, h_K1, h_K2, h_samples, S, 0, 1.0, nullptr, nullptr, nullptr, nullptr, h_E_out);
}

int clc_pnp_ransac(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const int32_t* h_samples,
                   int S, uint64_t seed, double thr2, double* h_Rt, uint8_t* h_inlier_mask, int* n_inliers, double* cost)
{
    return pnp_ransac_impl(ctx, h_X, h_x, N, h_K, h_samples, S, seed, thr2, h_Rt, h_inlier_mask, n_inliers, cost, nullptr);
}

int clc_pnp_refine(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const uint8_t* h_inlier_mask,
                   const double* h_Rt_in, double huber_a, int max_iter, double* h_Rt_out, double* h_cov, double* rmse, int* iterations)
{
    if (!ctx || N < 0 || !h_K || !h_Rt_in || (N > 0 && (!h_X || !h_x))) return fail(ctx, CLC_ERR_BAD_ARG, "pnp_refine: bad argument");
    if (N < 3) return fail(ctx, CLC_ERR_BAD_ARG, "pnp_refine: needs at least 3 correspondences");
    if (!(huber_a > 0.0)) huber_a = 16.0;          // ceres::HuberLoss(Square(4.0)), Refiner.hpp:122
    if (max_iter <= 0) max_iter = 50;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    // workspace in doubles: [ X 3N | x 2N | K 16 | Rt_in 12 | mask (N bytes) ] in, [ RefineOut ] out
    const size_t mask_d = ((size_t)N + 7) / 8;
    const size_t in_d = (size_t)5 * N + 16 + 12 + mask_d;
    const size_t out_d = (pnp_refine_out_bytes() + 7) / 8;
    int rc = ensure_pnp(ctx, in_d + out_d + 8);
    if (rc != CLC_OK) return rc;
    rc = ensure_pinned(ctx, (in_d > out_d ? in_d : out_d) * sizeof(double) + 64);
    if (rc != CLC_OK) return rc;
    double* dX = ctx->d_pnp;
    double* dx = dX + (size_t)3 * N;
    double* dK = dx + (size_t)2 * N;
    double* dRt = dK + 16;
    uint8_t* dMask = (uint8_t*)(dRt + 12);
    double* dOut = ctx->d_pnp + in_d;
    double* hp = (double*)ctx->h_pin;
    memcpy(hp, h_X, sizeof(double) * 3 * N);
    memcpy(hp + (size_t)3 * N, h_x, sizeof(double) * 2 * N);
    memcpy(hp + (size_t)5 * N, h_K, sizeof(double) * 9);
    memcpy(hp + (size_t)5 * N + 16, h_Rt_in, sizeof(double) * 12);
    if (h_inlier_mask) memcpy(hp + (size_t)5 * N + 28, h_inlier_mask, (size_t)N);
    CLC_HIP(ctx, hipMemcpyAsync(dX, hp, in_d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, launch_pnp_refine(dRt, dX, dx, h_inlier_mask ? dMask : nullptr, N, dK, huber_a, max_iter, dOut, ctx->stream, &ctx->prof));
    CLC_HIP(ctx, hipMemcpyAsync(hp, dOut, out_d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    struct { double Rt[12]; double cov[36]; double cost; double rmse; int32_t iterations; int32_t n_used; } r;
    memcpy(&r, hp, sizeof r);
    if (h_Rt_out) memcpy(h_Rt_out, r.Rt, sizeof r.Rt);
    if (h_cov) memcpy(h_cov, r.cov, sizeof r.cov);
    if (rmse) *rmse = r.rmse;
    if (iterations) *iterations = r.iterations;
    return CLC_OK;
}

int clc_pnp_localize(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const int32_t* h_samples,
                     int S, uint64_t seed, double thr2, double huber_a, double* h_Rt, double* h_cov, uint8_t* h_inlier_mask,
                     int* n_inliers, double* rmse)
{
    if (h_Rt) memset(h_Rt, 0, sizeof(double) * 12);
    if (h_cov) memset(h_cov, 0, sizeof(double) * 36);
    if (rmse) *rmse = 0.0;
    return pnp_ransac_impl(ctx, h_X, h_x, N, h_K, h_samples, S, seed, thr2, h_Rt, h_inlier_mask, n_inliers, nullptr, nullptr,
                           huber_a > 0.0 ? huber_a : 16.0, h_cov, rmse);
}

int clc_pnp_p3p(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const int32_t* h_samples, int S,
                double* h_Rt_out)
{
    if (!h_samples || !h_Rt_out) return fail(ctx, CLC_ERR_BAD_ARG, "pnp_p3p: bad argument");
    return pnp_ransac_impl(ctx, h_X, h_x, N, h_K, h_samples, S, 0, 1.0, nullptr, nullptr, nullptr, nullptr, h_Rt_out);
}

} // extern "C"

/* ---- a-contrario RANSAC ---------------------------------------------------------------------------------------------- */

namespace {

// log10 C(n, k) and log10 C(k, m), k = 0..n, as OpenMVG tabulates them: a FLOAT log10 table, float accumulation
// (logcombi<float>); the O(n^2) re-summation of the prefix is replaced by the running prefix -- same additions, same order.
// `lg` = the context's table of (float) log10(k), grown on demand and kept: it does not depend on n, and n + 1 calls of the
// portable log10 per solve were 40 us of a 220 us solve at n = 1000 (200 us at n = 5000).
void acr_tables(int n, int m, float* logc_n, float* logc_k, std::vector<float>& lg)
{
    if (lg.size() < (size_t)n + 2) {
        const size_t have = lg.size() < 1 ? 1 : lg.size();
        lg.resize((size_t)n + 2, 0.0f);
        lg[0] = 0.0f;
        for (size_t k = have; k < lg.size(); ++k) lg[k] = (float)clc_acr_log10((double)k);
    }
    std::vector<float> prefix((size_t)n + 1, 0.0f);
    for (int i = 1; i <= n; ++i) prefix[i] = prefix[i - 1] + (lg[n - i + 1] - lg[i]);
    for (int k = 0; k <= n; ++k) {
        uint32_t kk = (uint32_t)k;
        if (kk >= (uint32_t)n || kk == 0) { logc_n[k] = 0.0f; continue; }
        if ((uint32_t)n - kk < kk) kk = (uint32_t)n - kk;
        logc_n[k] = prefix[kk];
    }
    for (int nn = 0; nn <= n; ++nn) {
        uint32_t kk = (uint32_t)m;
        float r = 0.0f;
        if (!(kk >= (uint32_t)nn || kk == 0)) {
            if ((uint32_t)nn - kk < kk) kk = (uint32_t)nn - kk;
            for (uint32_t i = 1; i <= kk; ++i) r += lg[nn - i + 1] - lg[i];
        }
        logc_k[nn] = r;
    }
}

size_t dbl(size_t bytes) { return (bytes + 7) / 8; }

// kind 0: a = X (3 N), b = x (2 N), K1 = intrinsics; kind 1: a = x1, b = x2 (2 N each), K1 / K2, image 2 of img_w x img_h.
// h_model: 12 doubles [R|t] (kind 0) or {E (9), F (9)} (kind 1).
//
// One a-contrario solve as a small state machine (round 4): begin() stages the inputs and enqueues the first two rounds, poll() looks at the
// pinned progress word ONCE -- if the round the host waits for has come out it enqueues the next one (rounds stay enqueued one ahead of
// what the host knows) or moves on to the refinement, whose record it then polls the same way --, finish() copies the result out.  A
// single solve spins on poll() exactly as the loop it replaces did; clc_pnp_localize_ac_batch drives SEVERAL solves, each on a context of
// its own, from one thread: a solve is a chain of short launches with the host in the loop and leaves the GPU idle most of the time, so
// the chains of independent cameras interleave (BASELINE config[2]: "batched PnP/RANSAC pose").
struct AcrRun {
    // arguments
    clc_ctx* ctx = nullptr;
    int kind = 0, N = 0, img_w = 0, img_h = 0, max_iteration = 0;
    const double *h_a = nullptr, *h_b = nullptr, *h_K1 = nullptr, *h_K2 = nullptr;
    uint64_t seed = 0;
    double precision = 0.0, refine_huber = -1.0;
    double* h_model = nullptr; uint8_t* h_mask = nullptr; int32_t* h_inliers = nullptr;
    int *n_inliers = nullptr, *iterations = nullptr, *rounds = nullptr;
    double *error_max = nullptr, *min_nfa = nullptr, *h_cov = nullptr, *rmse = nullptr;
    // state
    enum Phase { IDLE, ROUNDS, REFINE, DONE } phase = IDLE;
    int status = CLC_OK;
    int m = 0, M = 0, md = 0;
    bool refine = false;
    AcrProblem pb{};
    hipStream_t st = nullptr;
    int launches = 0, bound = 0, reserve0 = 0;
    uint32_t round = 0, spins = 0;
    std::chrono::steady_clock::time_point wait_start;
    // lockstep batches (drive_group): the solve's rounds ride in launches shared with the other solves of the batch, on group_stream; the
    // run only watches its word -- `reported` = the round waited for has come out, `more` = it needs another one (bound: its batch bound)
    bool grouped = false, reported = false, more = false;
    hipStream_t group_stream = nullptr, refine_st = nullptr;
    int first_bound = 0;
    const double* stage_src = nullptr; double* stage_dst = nullptr; size_t stage_n = 0;   // the inputs' way to the device (a launch, not a copy)
    int batch_cap = kAcrMaxBatch;      // most iterations a round evaluates (CLC_ACR_BATCH_CAP; batches: see acr_batch_cap)
    unsigned long long* h_word = nullptr;
    AcrResult* h_res = nullptr;
    int32_t* p_inl = nullptr;
    double* p_ref = nullptr;
    int32_t* ready = nullptr;
    double *d_a = nullptr, *d_b = nullptr, *d_K1 = nullptr, *d_K2 = nullptr, *d_models = nullptr, *d_ref = nullptr;
    AcrState* d_state = nullptr; AcrHyp* d_hyp = nullptr;
    uint32_t *d_sorted = nullptr, *d_best = nullptr, *d_index = nullptr;
    AcrResult* d_res = nullptr; uint8_t* d_mask = nullptr;

    // Every failure after the first launch drains the stream first (ignoring what the drain reports): launches of the failed solve may
    // still be in flight and would otherwise write the progress word / result record of the NEXT solve, which reuses the same pinned block.
    int drained(const int code) { (void)hipStreamSynchronize(st); phase = DONE; status = code; return code; }
    int stop(const int code) { phase = DONE; status = code; return code; }

    int enqueue_round(const int bnd)
    {
        const int S = bnd < 1 ? 1 : (bnd > kAcrMaxBatch ? kAcrMaxBatch : bnd);
        if (kind == 0) {
            // one launch: replay of the previous round, this round's samples, P3P, residuals / sort / NFA; the word of round r comes
            // out of launch r + 1
            CLC_HIP(ctx, launch_acr_round_p3p(pb, launches & 1, d_state, d_hyp, d_sorted, d_models, d_best, d_index, h_word, st, S, d_mask,
                                              d_res, nullptr, p_inl, h_res));
        } else {
            // two launches: replay of the previous round + this round's samples + five-point solve, then nfa; the word of round r
            // comes out of round r + 1's first launch
            CLC_HIP(ctx, launch_acr_round_5pt(pb, launches & 1, d_state, d_hyp, d_sorted, d_models, d_best, d_index, h_word, st, S, d_mask,
                                              d_res, nullptr, p_inl, h_res));
        }
        ++launches;
        return CLC_OK;
    }

    // validates, stages, enqueues the first two rounds.  Returns a status; phase == DONE afterwards means there is nothing to wait for.
    int begin()
    {
        m = kind == 0 ? 3 : 5; M = kind == 0 ? 4 : 10; md = kind == 0 ? 12 : 18;
        const int ad = kind == 0 ? 3 : 2;
        if (!ctx || N < 0 || max_iteration < 0 || !h_K1 || (kind == 1 && !h_K2) || (N > 0 && (!h_a || !h_b)))
            return stop(fail(ctx, CLC_ERR_BAD_ARG, "acransac: bad argument"));
        if (n_inliers) *n_inliers = 0;
        if (error_max) *error_max = 0.0;
        if (min_nfa) *min_nfa = INFINITY;
        if (iterations) *iterations = 0;
        if (rounds) *rounds = 0;
        if (h_mask && N > 0) memset(h_mask, 0, (size_t)N);
        if (N <= m || max_iteration == 0) return stop(CLC_OK);                 // ACRANSAC: nData <= sizeSample -> (0, 0), no model
        if (N > kAcrMaxN) return stop(fail(ctx, CLC_ERR_CAPACITY, "acransac: more than 16384 correspondences per solve"));
        if (max_iteration > 500000) return stop(fail(ctx, CLC_ERR_CAPACITY, "acransac: more than 500000 iterations"));
        if (kind == 1 && (img_w <= 0 || img_h <= 0)) return stop(fail(ctx, CLC_ERR_BAD_ARG, "acransac: image size needed for the point-to-line model"));
        phase = DONE; status = CLC_ERR_HIP;                                     // (what an early CLC_HIP return leaves behind)
        const int rc0 = begin_body(ad);
        if (rc0 != CLC_OK) { phase = DONE; status = rc0; }
        return rc0;
    }

    int begin_body(const int ad)
    {
        CLC_HIP(ctx, hipSetDevice(ctx->device));
        refine = kind == 0 && refine_huber > 0.0;
        // device workspace (doubles): [ a | b | K1 16 | K2 16 | logc_n | logc_k | initial state ] staged by
        // one launch, then scratch
        const size_t state_d = 2 * dbl(sizeof(AcrState));                      // [ copy 0 | copy 1 = the state a run starts from ]
        const size_t in_d = (size_t)(ad + 2) * N + 32 + 2 * dbl(sizeof(float) * ((size_t)N + 1)) + state_d;
        // a round's launches read what the round before them wrote: two copies of state, models, slots and sorted lists, indexed by
        // launch parity (acransac.hip: acr_round_kernel, acr_solve5_kernel)
        if (grouped) batch_cap = kind == 0 ? 8 : 12;                   // (a shared launch carries every chain's speculative slots: shorter rounds)
        if (const char* e = getenv("CLC_ACR_BATCH_CAP")) { const int v = atoi(e); if (v >= 1 && v <= kAcrMaxBatch) batch_cap = v; }
        const int copies = 2;
        const size_t models_d = (size_t)copies * kAcrMaxBatch * M * md;
        const size_t hyp_d = dbl(acr_hyp_bytes() * copies * kAcrMaxBatch * M);
        const size_t sorted_d = dbl(sizeof(uint32_t) * (size_t)copies * kAcrMaxBatch * M * N);
        const size_t idx_d = dbl(sizeof(uint32_t) * (size_t)N);
        const size_t res_d = dbl(sizeof(AcrResult)), mask_d = dbl((size_t)N);
        const size_t ref_d = refine ? dbl(pnp_refine_out_bytes()) : 0;
        int rc = ensure_pnp(ctx, in_d + models_d + hyp_d + sorted_d + 2 * idx_d + res_d + mask_d + ref_d + 16);
        if (rc != CLC_OK) return rc;
        // pinned: [ inputs | state mirror | sequence word | result | mask | inlier list | refine record ]
        const size_t inl_d = dbl(sizeof(int32_t) * (size_t)N);
        rc = ensure_pinned(ctx, (in_d + state_d + 1 + res_d + mask_d + inl_d + ref_d) * sizeof(double) + 64);   // (+1: the polled word)
        if (rc != CLC_OK) return rc;
        double* d = ctx->d_pnp;
        d_a = d;                               d += (size_t)ad * N;
        d_b = d;                               d += (size_t)2 * N;
        d_K1 = d;                              d += 16;
        d_K2 = d;                              d += 16;
        float* d_cn = (float*)d;               d += dbl(sizeof(float) * ((size_t)N + 1));
        float* d_ck = (float*)d;               d += dbl(sizeof(float) * ((size_t)N + 1));
        d_state = (AcrState*)d;                d += state_d;
        d_models = d;                          d += models_d;
        d_hyp = (AcrHyp*)d;                    d += hyp_d;
        d_sorted = (uint32_t*)d;               d += sorted_d;
        d_best = (uint32_t*)d;                 d += idx_d;
        d_index = (uint32_t*)d;                d += idx_d;
        d_res = (AcrResult*)d;                 d += res_d;
        d_mask = (uint8_t*)d;                  d += mask_d;
        d_ref = d;
        double* hp = (double*)ctx->h_pin;
        memcpy(hp, h_a, sizeof(double) * ad * N);
        memcpy(hp + (size_t)ad * N, h_b, sizeof(double) * 2 * N);
        double* hK = hp + (size_t)(ad + 2) * N;
        memset(hK, 0, sizeof(double) * 32);
        memcpy(hK, h_K1, sizeof(double) * 9);
        if (h_K2) memcpy(hK + 16, h_K2, sizeof(double) * 9);
        float* h_cn = (float*)(hK + 32);
        float* h_ck = (float*)(hK + 32 + dbl(sizeof(float) * ((size_t)N + 1)));
        acr_tables(N, m, h_cn, h_ck, ctx->acr_lg);
        // the state ACRANSAC starts from is part of the upload (every round draws its own samples on the device)
        AcrState* h_states = (AcrState*)(hK + 32 + 2 * dbl(sizeof(float) * ((size_t)N + 1)));
        memset(h_states, 0, 2 * sizeof(AcrState));
        // launch 0 (parity 0) reads copy 1
        AcrState* h_init = h_states + 1;
        h_init->min_nfa = INFINITY; h_init->error_max = INFINITY;
        h_init->best_iter = -1;
        h_init->reserve = max_iteration / 10;
        h_init->n_iter = max_iteration - h_init->reserve;
        h_init->n_index = N; h_init->index_all = 1;
        h_init->ac_mode = std::isinf(precision) ? 1 : 0;
        h_init->grow = batch_cap < 32 ? batch_cap : 32;
        h_init->cur_batch = h_init->n_iter < h_init->grow ? h_init->n_iter : h_init->grow;
        h_word = (unsigned long long*)(hp + in_d + state_d);
        h_res = (AcrResult*)(hp + in_d + state_d + 1);
        p_inl = (int32_t*)(hp + in_d + state_d + 1 + res_d + mask_d);
        p_ref = hp + in_d + state_d + 1 + res_d + mask_d + inl_d;
        __atomic_store_n(h_word, 0ull, __ATOMIC_RELAXED);

        pb = AcrProblem{};
        pb.kind = kind; pb.n = N; pb.m = m; pb.max_models = M; pb.model_doubles = md; pb.batch_cap = batch_cap;
        pb.a = d_a; pb.b = d_b; pb.K1 = d_K1; pb.K2 = d_K2; pb.logc_n = d_cn; pb.logc_k = d_ck;
        pb.loge0 = clc_acr_log10((double)M * (double)(N - m));
        if (kind == 0) {
            // ACKernelAdaptorResection_Intrinsics: residuals on the normalised camera plane (x 1 / focal), logalpha0 = log10(pi)
            pb.logalpha0 = clc_acr_log10(M_PI);
            pb.mult = 1.0;
            pb.norm = 1.0 / h_K1[0];
            for (int e = 0; e < 9; ++e) pb.K1v[e] = h_K1[e];
        } else {
            // ACKernelAdaptorEssential: point-to-line, logalpha0 = log10(2 D / A * 0.5) of image 2, error^(1/2)
            const double D = sqrt((double)img_w * (double)img_w + (double)img_h * (double)img_h), A = (double)img_w * (double)img_h;
            pb.logalpha0 = clc_acr_log10(2.0 * D / A * .5);
            pb.mult = 0.5;
            pb.norm = 1.0;
        }
        pb.max_threshold = std::isinf(precision) ? INFINITY : precision * (pb.norm * pb.norm);
        pb.seed = seed;

        st = grouped ? group_stream : ctx->stream;
        stage_src = hp; stage_dst = ctx->d_pnp; stage_n = (in_d + 1) & ~(size_t)1;      // (both blocks are sized past in_d + 1)
        if (!grouped) CLC_HIP(ctx, launch_acr_stage(stage_src, stage_dst, stage_n, st));  // (grouped: one launch for the batch, drive_group)
        if (!grouped) prof_mark(&ctx->prof, CLC_KERNEL_PNP_SCORE, true, st);
        // Rounds are enqueued ONE AHEAD of what the host knows: the solve / nfa / select kernels take the round's batch from the
        // device state (a round enqueued after the run has finished finds nothing to do), so the GPU goes from one round's
        // select straight into the next round's solve while the host is still polling (a 10 us bubble per round otherwise).
        launches = 0;
        // Upper bound of the batch a round can ask for, from what the host knows when it enqueues it (one or two rounds behind the
        // device): while the index set has not switched the batch doubles up to kAcrMaxBatch; afterwards it is what is left of the
        // reserve, remaining = n_iter - iter (+ a margin for the "no inliers: n_iter++" rule, once per round).
        reserve0 = h_init->reserve;
        bound = h_init->n_iter < batch_cap ? h_init->n_iter : batch_cap;
        first_bound = h_init->cur_batch;                               // (a replaying launch takes its first batch as it stands)
        if (!grouped) {
            int rc2 = enqueue_round(first_bound);
            if (rc2 != CLC_OK) return drained(rc2);
            rc2 = enqueue_round(bound);                                 // speculative: the round after the one being waited for
            if (rc2 != CLC_OK) return drained(rc2);
        }                                                               // (grouped: drive_group enqueues the shared launches)
        round = 1;
        spins = 0;
        reported = false; more = false;
        wait_start = std::chrono::steady_clock::now();
        phase = ROUNDS;
        status = CLC_OK;
        return CLC_OK;
    }

    // this solve's part of a shared launch
    void chain(AcrChain& c) const
    {
        c.pb = pb;
        c.states = d_state; c.hyps = d_hyp; c.sorted = d_sorted; c.models = d_models; c.best_inliers = d_best; c.index_set = d_index;
        c.h_word = h_word;
        c.fin = AcrFinish{ d_mask, d_res, nullptr, p_inl, h_res };
    }
    // the shared launch of the next round is in the stream: wait for that round's word
    void advance()
    {
        ++round; spins = 0; reported = false; more = false;
        wait_start = std::chrono::steady_clock::now();
    }

    // One look at the progress word / the refinement's ready flag.  Returns the status; phase == DONE when the solve has ended.
    int poll()
    {
        if (phase == ROUNDS) {
            // the select kernel publishes one packed word (round number, iterations consumed, iter, n_iter) in pinned memory: poll it
            // (a stream synchronisation costs ~10 us per round); after 2 ms without progress fall back to the synchronisation, which
            // also surfaces errors
            unsigned long long w = __atomic_load_n(h_word, __ATOMIC_ACQUIRE);
            if ((w >> 49) < (round & 0x7FFFu)) {
                if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - wait_start > std::chrono::milliseconds(2)) {
                    const hipError_t e = hipStreamSynchronize(st);
                    if (e != hipSuccess) return drained(fail(ctx, CLC_ERR_HIP, "hipStreamSynchronize(st)", e));
                    w = __atomic_load_n(h_word, __ATOMIC_ACQUIRE);
                    if ((w >> 49) < (round & 0x7FFFu)) return drained(fail(ctx, CLC_ERR_HIP, "acransac: round did not complete"));
                } else return CLC_OK;
            }
            const int iter_k = (int)(w & 0xFFFFFu), n_iter_k = (int)((w >> 20) & 0xFFFFFu);
            if (iter_k < n_iter_k) {
                const bool switched = ((w >> 48) & 1u) != 0;
                const long left = (long)n_iter_k - iter_k + 4 + (switched ? 0 : reserve0);
                bound = left > batch_cap ? batch_cap : (int)left;
                if (round > 0x7000u) return drained(fail(ctx, CLC_ERR_STATE, "acransac: too many rounds"));
                if (grouped) { reported = true; more = true; return CLC_OK; }      // (drive_group enqueues the batch's next launch)
                const int rc = enqueue_round(bound);                   // speculative: the round after the one being waited for
                if (rc != CLC_OK) return drained(rc);
                ++round;
                spins = 0;
                wait_start = std::chrono::steady_clock::now();
                return CLC_OK;
            }
            // done: the completing round has left the result in pinned memory
            reported = true; more = false;
            if (!grouped) prof_mark(&ctx->prof, CLC_KERNEL_PNP_SCORE, false, st);
            // The result record, mask and inlier list were written by the round that completed the run BEFORE its word (system-scope
            // release / acquire): no finish launch, and without refinement no stream synchronisation either -- the round enqueued ahead is
            // still in the stream, evaluates nothing and touches no host memory; anything enqueued later on this stream is ordered behind it.
            if (!refine) { phase = DONE; return CLC_OK; }
            // Behind the launch that completed the run, on the same stream (nothing is queued behind that launch any more: the word of
            // the last round comes out of the last launch).  The refinement writes its record into pinned memory and sets `ready` last;
            // the host polls that instead of synchronising the stream (~5 us), with the synchronisation as the fallback after 5 ms.
            // (A grouped run refines on its OWN context's stream: the shared stream still carries the other solves' rounds.  What the
            // refinement reads was written before the word the host has just seen -- system-scope release / acquire -- so the launch
            // needs no ordering against the shared stream.)
            ready = (int32_t*)((uint8_t*)p_ref + pnp_refine_ready_offset());
            __atomic_store_n(ready, 0, __ATOMIC_RELAXED);
            refine_st = grouped ? ctx->stream : st;
            const hipError_t e = launch_pnp_refine((const double*)d_res /* AcrResult.model = [R|t] */, d_a, d_b, d_mask, N, d_K1, refine_huber, 50,
                                                   d_ref, refine_st, &ctx->prof, &d_res->valid, p_ref);
            if (e != hipSuccess) return drained(fail(ctx, CLC_ERR_HIP, "launch_pnp_refine", e));
            spins = 0;
            wait_start = std::chrono::steady_clock::now();
            phase = REFINE;
            return CLC_OK;
        }
        if (phase == REFINE) {
            if (__atomic_load_n(ready, __ATOMIC_ACQUIRE) == 0) {
                if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - wait_start > std::chrono::milliseconds(5)) {
                    const hipError_t e = hipStreamSynchronize(refine_st);
                    if (e != hipSuccess) return drained(fail(ctx, CLC_ERR_HIP, "hipStreamSynchronize(refine stream)", e));
                    if (__atomic_load_n(ready, __ATOMIC_ACQUIRE) == 0) return drained(fail(ctx, CLC_ERR_HIP, "acransac: refinement did not complete"));
                } else return CLC_OK;
            }
            phase = DONE;
        }
        return status;
    }

    // after phase == DONE with status CLC_OK and a run that was started: the result into the caller's buffers
    void finish()
    {
        if (status != CLC_OK || !h_res) return;
        const AcrResult r = *h_res;
        if (h_model) {
            if (kind == 0) memcpy(h_model, r.model, sizeof(double) * 12);
            else { memcpy(h_model, r.model + 9, sizeof(double) * 9); memcpy(h_model + 9, r.model, sizeof(double) * 9); }   // slots hold {F, E}
        }
        // the mask is rebuilt from the inlier list here (h_mask was cleared above): the device does not push N bytes + one scattered byte
        // per inlier over PCIe for it
        if (h_mask) for (int i = 0; i < r.n_inliers; ++i) h_mask[p_inl[i]] = 1;
        if (h_inliers && r.n_inliers > 0) memcpy(h_inliers, p_inl, sizeof(int32_t) * (size_t)r.n_inliers);
        if (n_inliers) *n_inliers = r.n_inliers;
        if (error_max) *error_max = r.error_max;
        if (min_nfa) *min_nfa = r.min_nfa;
        if (iterations) *iterations = r.iterations;
        if (rounds) *rounds = r.rounds;
        if (refine) {
            struct { double Rt[12]; double cov[36]; double cost; double rmse; int32_t iterations; int32_t n_used; } f;
            memcpy(&f, p_ref, sizeof f);
            if (r.n_inliers > 0) {
                if (h_model) memcpy(h_model, f.Rt, sizeof f.Rt);
                if (h_cov) memcpy(h_cov, f.cov, sizeof f.cov);
                if (rmse) *rmse = f.rmse;
            }
        }
    }
};

// All runs to their end from ONE host thread: whichever solve's round has come out gets its next one enqueued (a run is a chain of short
// launches with the host in the loop, so several runs interleave on the device).  Round 5 measured two and three driving threads (the
// runs dealt out, each thread on its runs' own contexts and streams): eight two-view filters 0.134 -> 0.136 ms per pair, eight poses
// 0.049 -> 0.052 ms per pose -- the chains' own latency, not the host's launch calls, is what a batch waits for; one thread stays.
void drive_runs(std::vector<AcrRun>& runs)
{
    size_t live = 0;
    for (AcrRun& r : runs) if (r.phase != AcrRun::DONE) ++live;
    while (live > 0)
        for (AcrRun& r : runs) {
            if (r.phase == AcrRun::DONE) continue;
            (void)r.poll();
            if (r.phase == AcrRun::DONE) --live;
        }
}

// Lockstep form of the same (round 5, the default of the batched entries): the batch's solves -- one kind, contexts on one device -- share
// their launches.  Round r of every unfinished solve is ONE launch (resection) or two (two-view) with blockIdx.y = solve
// (launch_acr_round_*_chains), on the first context's stream, enqueued one ahead as for a single solve; the host waits for the round's
// words of all unfinished solves and enqueues the next shared launch while any of them needs one.  A finished solve's part of the later
// launches finds nothing to replay and returns.  Why: eight interleaved poses were ~80 launches from one thread (4-5 us each inside the
// runtime, and the runtime serialises launching threads), eight two-view filters ~100; in lockstep they are ~10 and ~16.  Same bits per
// solve: a chain's kernels take its batch from its own device state, the grid and the sort width (the largest chain's) only bound them.
// Measured (MI355X, N = 1 000, 30 % outliers, tools/time_two_view.py / time_pose_batch2.py; interleaved -> lockstep, one staging launch
// for the batch, a round's launches carrying only the solves still in their rounds):
//     two-view filters   2: 0.417 -> 0.439 ms   4: 0.570 -> 0.574   8: 1.08-1.21 -> 0.75-0.80      (rounds of <= 12 / 16 iterations)
//     resection poses    2: 0.190 -> 0.189      4: 0.251 -> 0.271   8: 0.514 -> 0.508; with rounds of <= 8 iterations 0.429 (4: 0.295)
// The interleaved batch is bound by the host's launch calls (eight poses = ~70 launches of 5-7 us from one thread, two of them in flight on
// the device at a time); in lockstep eight poses are ~12 launches, but each carries every solve's speculative slots -- 8 x 32 iterations x 4
// models of 1 024 threads do not fit the chip at once (33 us per round against 17.5) -- hence the cap on a round's iterations, which costs
// rounds.  The two-view round is a 44 us chain of dependent fp64 steps in ONE wave per iteration: eight chains' solves in one launch cost
// what one costs.  Default: lockstep for two-view batches of four or more (rounds of <= 12 iterations) and resection batches of eight or
// more (<= 8); CLC_ACR_LOCKSTEP=1 / =0 forces it on (any batch of two or more) / off for both kinds, CLC_ACR_BATCH_CAP the iterations.
bool acr_lockstep(const int kind, const int n_jobs)
{
    static const int mode = [] { const char* e = getenv("CLC_ACR_LOCKSTEP"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
    if (n_jobs < 2 || mode == 0) return false;
    if (mode == 1) return true;
    return kind == 1 ? n_jobs >= 4 : n_jobs >= 8;
}
void drive_group(std::vector<AcrRun>& runs)
{
    std::vector<AcrRun*> live;
    for (AcrRun& r : runs) if (r.phase == AcrRun::ROUNDS && r.grouped) live.push_back(&r);
    if (live.empty()) { drive_runs(runs); return; }
    const int kind = live[0]->kind;
    hipStream_t st = live[0]->group_stream;
    clc_ctx* ctx0 = live[0]->ctx;
    int launches = 0;
    auto fail_all = [&](const int code) { for (AcrRun* r : live) if (r->phase != AcrRun::DONE) (void)r->drained(code); };
    if (hipSetDevice(ctx0->device) != hipSuccess) { fail_all(fail(ctx0, CLC_ERR_HIP, "hipSetDevice")); return; }
    // the inputs of all solves: one staging launch per kMaxBatch of them
    for (size_t k = 0; k < live.size(); k += kMaxBatch) {
        const int n = (int)std::min<size_t>(kMaxBatch, live.size() - k);
        const double* src[kMaxBatch]; double* dst[kMaxBatch]; size_t cnt[kMaxBatch];
        for (int i = 0; i < n; ++i) { src[i] = live[k + i]->stage_src; dst[i] = live[k + i]->stage_dst; cnt[i] = live[k + i]->stage_n; }
        const hipError_t e = launch_acr_stage_chains(src, dst, cnt, n, st);
        if (e != hipSuccess) { fail_all(fail(ctx0, CLC_ERR_HIP, "acransac: shared staging launch", e)); return; }
    }
    // a round's launches carry the solves that are still in their rounds (what the host knows when it enqueues: a solve that finishes in
    // the round being waited for rides along once more and finds nothing to replay), kMaxBatch per launch
    auto enqueue = [&](const int bound) -> bool {
        AcrChains pack;
        int n = 0;
        auto flush = [&]() -> bool {
            if (n == 0) return true;
            const hipError_t e = kind == 0 ? launch_acr_round_p3p_chains(pack, n, launches & 1, bound, st)
                                           : launch_acr_round_5pt_chains(pack, n, launches & 1, bound, st);
            n = 0;
            if (e != hipSuccess) { fail_all(fail(ctx0, CLC_ERR_HIP, "acransac: shared round launch", e)); return false; }
            return true;
        };
        for (AcrRun* r : live) {
            if (r->phase != AcrRun::ROUNDS) continue;
            r->chain(pack.c[n++]);
            if (n == kMaxBatch && !flush()) return false;
        }
        if (!flush()) return false;
        ++launches;
        return true;
    };
    int b0 = 1, b1 = 1;
    for (AcrRun* r : live) { b0 = std::max(b0, r->first_bound); b1 = std::max(b1, r->bound); }
    if (!enqueue(b0) || !enqueue(b1)) return;                        // the second: speculative, the round after the one being waited for
    for (;;) {
        bool any_live = false, rounds_waiting = false, any_more = false;
        int bnd = 1;
        for (AcrRun* r : live) {
            if (r->phase == AcrRun::DONE) continue;
            any_live = true;
            if (!(r->phase == AcrRun::ROUNDS && r->reported)) (void)r->poll();   // ROUNDS: one look at the word; REFINE: one look at the record
            if (r->phase != AcrRun::ROUNDS) continue;
            if (!r->reported) rounds_waiting = true;                  // (in its rounds and reported = it needs another round)
            else { any_more = true; bnd = std::max(bnd, r->bound); }
        }
        if (!any_live) break;
        if (any_more && !rounds_waiting) {
            // every solve still in its rounds has reported the round waited for: the next shared launch (refinements of finished solves
            // may still be out on their own streams; they do not hold the rounds up)
            if (!enqueue(bnd)) return;
            for (AcrRun* r : live) if (r->phase == AcrRun::ROUNDS) r->advance();
        }
    }
    // The launch enqueued ahead of the last round is still in the shared stream (it finds nothing to replay, but its keepers carry every
    // chain's state forward inside that chain's workspace): whatever the other contexts enqueue next on THEIR streams -- the staging of
    // their next solve, a refinement -- has to come behind it.  (Work on the shared stream is ordered by the stream.)
    bool ordered = false;
    if (ctx0->ev_group || hipEventCreateWithFlags(&ctx0->ev_group, hipEventDisableTiming) == hipSuccess) {
        ordered = hipEventRecord(ctx0->ev_group, st) == hipSuccess;
        for (AcrRun* r : live)
            if (ordered && r->ctx->stream != st) ordered = hipStreamWaitEvent(r->ctx->stream, ctx0->ev_group, 0) == hipSuccess;
    }
    if (!ordered) (void)hipStreamSynchronize(st);
    // (runs of the batch that were not part of the group -- early outs -- are DONE already)
}

int acr_impl(clc_ctx* ctx, int kind, const double* h_a, const double* h_b, int N, const double* h_K1, const double* h_K2, int img_w,
             int img_h, int max_iteration, uint64_t seed, double precision, double refine_huber, double* h_model, uint8_t* h_mask,
             int32_t* h_inliers, int* n_inliers, double* error_max, double* min_nfa, int* iterations, int* rounds, double* h_cov,
             double* rmse)
{
    AcrRun run;
    run.ctx = ctx; run.kind = kind; run.h_a = h_a; run.h_b = h_b; run.N = N; run.h_K1 = h_K1; run.h_K2 = h_K2; run.img_w = img_w; run.img_h = img_h;
    run.max_iteration = max_iteration; run.seed = seed; run.precision = precision; run.refine_huber = refine_huber;
    run.h_model = h_model; run.h_mask = h_mask; run.h_inliers = h_inliers; run.n_inliers = n_inliers; run.error_max = error_max;
    run.min_nfa = min_nfa; run.iterations = iterations; run.rounds = rounds; run.h_cov = h_cov; run.rmse = rmse;
    int rc = run.begin();
    if (rc != CLC_OK) return rc;
    while (run.phase != AcrRun::DONE) {
        rc = run.poll();
        if (rc != CLC_OK) return rc;
    }
    run.finish();
    return run.status;
}

} // namespace

extern "C" {

int clc_pnp_acransac(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, int max_iteration, uint64_t seed,
                     double precision, double* h_Rt, uint8_t* h_inlier_mask, int32_t* h_inliers, int* n_inliers, double* error_max,
                     double* min_nfa, int* iterations)
{
    if (h_Rt) memset(h_Rt, 0, sizeof(double) * 12);
    return acr_impl(ctx, 0, h_X, h_x, N, h_K, nullptr, 0, 0, max_iteration, seed, precision, -1.0, h_Rt, h_inlier_mask, h_inliers, n_inliers,
                    error_max, min_nfa, iterations, nullptr, nullptr, nullptr);
}

int clc_pnp_localize_ac(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, int max_iteration, uint64_t seed,
                        double precision, double huber_a, double* h_Rt, double* h_cov, uint8_t* h_inlier_mask, int32_t* h_inliers,
                        int* n_inliers, double* error_max, double* rmse)
{
    if (h_Rt) memset(h_Rt, 0, sizeof(double) * 12);
    if (h_cov) memset(h_cov, 0, sizeof(double) * 36);
    if (rmse) *rmse = 0.0;
    return acr_impl(ctx, 0, h_X, h_x, N, h_K, nullptr, 0, 0, max_iteration, seed, precision, huber_a > 0.0 ? huber_a : 16.0, h_Rt,
                    h_inlier_mask, h_inliers, n_inliers, error_max, nullptr, nullptr, nullptr, h_cov, rmse);
}

int clc_pnp_localize_ac_batch(clc_ctx* const* ctxs, clc_pose_job* jobs, int n_jobs)
{
    if (n_jobs < 0 || (n_jobs > 0 && (!ctxs || !jobs))) return CLC_ERR_BAD_ARG;
    if (n_jobs == 0) return CLC_OK;
    for (int i = 0; i < n_jobs; ++i) {
        if (!ctxs[i]) return CLC_ERR_BAD_ARG;
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(ctxs[i], CLC_ERR_BAD_ARG, "pnp_localize_ac_batch: every job needs a context of its own");
        if (ctxs[i]->device != ctxs[0]->device) return fail(ctxs[i], CLC_ERR_BAD_ARG, "pnp_localize_ac_batch: the contexts must live on one device");
    }
    std::vector<AcrRun> runs((size_t)n_jobs);
    int worst = CLC_OK, live = 0;
    const bool lockstep = acr_lockstep(0, n_jobs);
    for (int i = 0; i < n_jobs; ++i) {
        clc_pose_job& jb = jobs[i];
        AcrRun& r = runs[(size_t)i];
        r.grouped = lockstep; r.group_stream = ctxs[0]->stream;
        if (jb.Rt) memset(jb.Rt, 0, sizeof(double) * 12);
        if (jb.cov) memset(jb.cov, 0, sizeof(double) * 36);
        jb.n_inliers = 0; jb.error_max = 0.0; jb.rmse = 0.0; jb.iterations = 0;
        r.ctx = ctxs[i]; r.kind = 0; r.h_a = jb.X; r.h_b = jb.x; r.N = jb.n; r.h_K1 = jb.K;
        r.max_iteration = jb.max_iteration; r.seed = jb.seed; r.precision = jb.precision;
        r.refine_huber = jb.refine ? (jb.huber_a > 0.0 ? jb.huber_a : 16.0) : -1.0;
        r.h_model = jb.Rt; r.h_mask = jb.inlier_mask; r.h_inliers = jb.inliers; r.n_inliers = &jb.n_inliers; r.error_max = &jb.error_max;
        r.iterations = &jb.iterations; r.h_cov = jb.cov; r.rmse = &jb.rmse;
        jb.status = r.begin();                        // stages this job's inputs (and, on its own, puts its first two rounds into its context's stream)
        if (r.phase != AcrRun::DONE) ++live;
    }
    (void)live;
    if (lockstep) drive_group(runs); else drive_runs(runs);
    for (int i = 0; i < n_jobs; ++i) {
        AcrRun& r = runs[(size_t)i];
        r.finish();
        jobs[i].status = r.status;
        if (r.status != CLC_OK && worst == CLC_OK) worst = r.status;
    }
    return worst;
}

int clc_essential_acransac(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1, const double* h_K2,
                           int img_w, int img_h, int max_iteration, uint64_t seed, double precision, double* h_E, double* h_F,
                           uint8_t* h_inlier_mask, int32_t* h_inliers, int* n_inliers, double* error_max, double* min_nfa, int* iterations)
{
    double EF[18] = {};
    const int rc = acr_impl(ctx, 1, h_x1, h_x2, N, h_K1, h_K2, img_w, img_h, max_iteration, seed, precision, -1.0, EF, h_inlier_mask,
                            h_inliers, n_inliers, error_max, min_nfa, iterations, nullptr, nullptr, nullptr);
    if (h_E) memcpy(h_E, EF, sizeof(double) * 9);
    if (h_F) memcpy(h_F, EF + 9, sizeof(double) * 9);
    return rc;
}

} // extern "C"

// ---- several two-view problems at once / the inter-camera step behind the C ABI (round 5) -----------------------------------
namespace {

void two_view_begin(AcrRun& r, clc_ctx* ctx, clc_two_view_job& jb, double* EF, const bool lockstep, hipStream_t group_stream)
{
    r.grouped = lockstep; r.group_stream = group_stream;
    if (jb.E) memset(jb.E, 0, sizeof(double) * 9);
    if (jb.F) memset(jb.F, 0, sizeof(double) * 9);
    jb.n_inliers = 0; jb.iterations = 0; jb.error_max = 0.0; jb.min_nfa = INFINITY;
    r.ctx = ctx; r.kind = 1; r.h_a = jb.x1; r.h_b = jb.x2; r.N = jb.n; r.h_K1 = jb.K1; r.h_K2 = jb.K2; r.img_w = jb.img_w; r.img_h = jb.img_h;
    r.max_iteration = jb.max_iteration; r.seed = jb.seed; r.precision = jb.precision; r.refine_huber = -1.0;
    r.h_model = EF; r.h_mask = jb.inlier_mask; r.h_inliers = jb.inliers; r.n_inliers = &jb.n_inliers; r.error_max = &jb.error_max;
    r.min_nfa = &jb.min_nfa; r.iterations = &jb.iterations;
    jb.status = r.begin();
}

int check_batch_contexts(clc_ctx* const* ctxs, int n_jobs, const char* what)
{
    for (int i = 0; i < n_jobs; ++i) {
        if (!ctxs[i]) return CLC_ERR_BAD_ARG;
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(ctxs[i], CLC_ERR_BAD_ARG, what);
        if (ctxs[i]->device != ctxs[0]->device) return fail(ctxs[i], CLC_ERR_BAD_ARG, "batch: the contexts must live on one device");
    }
    return CLC_OK;
}

// pixel -> normalised camera plane for an upper-triangular K (row-major)
inline void normalise_px(const double* K, const double x, const double y, double* n)
{
    n[1] = (y - K[5]) / K[4];
    n[0] = (x - K[2] - K[1] * n[1]) / K[0];
}

// The host part of the inter-camera step between the two-view filter and the refinement (coloc.hpp:296-340): relative pose from E with
// the chirality vote (RobustMatcher.hpp:176-183), the pair's temporary map in the source camera's frame, its scale against the global
// map through the features both hold (colocUtils.hpp:184-211: mean of consecutive distance ratios, behind a depth-ratio screen), the
// destination's first pose through the source's.  Triangulation: the depths along the two rays that bring them closest (closed form;
// OpenMVG's TriangulateDLT differs from it by less than the measurement noise).  Returns 0, or which stage failed (CLC_INTER_*).
int inter_geometry(clc_inter_pose_job& jb, std::vector<double>& Xw, std::vector<double>& x2f)
{
    const clc_two_view_job& tv = jb.tv;
    const int ni = tv.n_inliers;
    jb.n_front = 0; jb.n_common = 0; jb.scale = 0.0;
    if (ni < 13 || !tv.E || !tv.inliers) return CLC_INTER_NO_MODEL;
    openMVG::Mat3 E;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E(i, j) = tv.E[3 * i + j];
    std::vector<openMVG::geometry::Pose3> cand;
    coloc::hipgeom::motion_from_essential(E, &cand);
    std::vector<double> n1((size_t)2 * ni), n2((size_t)2 * ni);
    for (int k = 0; k < ni; ++k) {
        const int i = tv.inliers[k];
        normalise_px(tv.K1, tv.x1[2 * i], tv.x1[2 * i + 1], &n1[2 * (size_t)k]);
        normalise_px(tv.K2, tv.x2[2 * i], tv.x2[2 * i + 1], &n2[2 * (size_t)k]);
    }
    int best = -1, best_cnt = -1;
    std::vector<double> l1((size_t)ni), bl1;
    std::vector<uint8_t> fr((size_t)ni), bfr;
    double Rb[9] = {}, tb[3] = {};
    for (size_t c = 0; c < cand.size(); ++c) {
        const openMVG::Mat3& R = cand[c].rotation();
        const openMVG::Vec3 t = cand[c].translation();
        int cnt = 0;
        for (int k = 0; k < ni; ++k) {
            const double p[3] = { n1[2 * (size_t)k], n1[2 * (size_t)k + 1], 1.0 }, b[3] = { n2[2 * (size_t)k], n2[2 * (size_t)k + 1], 1.0 };
            double a[3];
            for (int r = 0; r < 3; ++r) a[r] = R(r, 0) * p[0] + R(r, 1) * p[1] + R(r, 2) * p[2];
            // min | l1 a - l2 b + t |^2
            const double aa = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], bb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
            const double ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
            const double at = a[0] * t[0] + a[1] * t[1] + a[2] * t[2], bt = b[0] * t[0] + b[1] * t[1] + b[2] * t[2];
            double det = aa * bb - ab * ab;
            if (std::fabs(det) < 1e-18) det = 1e-18;
            const double d1 = (-at * bb + bt * ab) / det, d2 = (-at * ab + bt * aa) / det;
            l1[(size_t)k] = d1;
            fr[(size_t)k] = d1 > 0.0 && d2 > 0.0;
            cnt += fr[(size_t)k];
        }
        if (cnt > best_cnt) {
            best_cnt = cnt; best = (int)c; bl1 = l1; bfr = fr;
            for (int r = 0; r < 3; ++r) { for (int q = 0; q < 3; ++q) Rb[3 * r + q] = R(r, q); tb[r] = t[r]; }
        }
    }
    if (best < 0 || best_cnt < 8) return CLC_INTER_NO_RELATIVE_POSE;
    jb.n_front = best_cnt;
    // the temporary map (source camera's frame, unit baseline) of the correspondences in front of both cameras
    const size_t nf = (size_t)best_cnt;
    std::vector<double> Xt(3 * nf);
    std::vector<int32_t> mi(nf);
    x2f.resize(2 * nf);
    size_t w = 0;
    for (int k = 0; k < ni; ++k) {
        if (!bfr[(size_t)k]) continue;
        const int i = tv.inliers[k];
        Xt[3 * w] = n1[2 * (size_t)k] * bl1[(size_t)k]; Xt[3 * w + 1] = n1[2 * (size_t)k + 1] * bl1[(size_t)k]; Xt[3 * w + 2] = bl1[(size_t)k];
        x2f[2 * w] = tv.x2[2 * i]; x2f[2 * w + 1] = tv.x2[2 * i + 1];
        const int32_t gi = jb.map_index ? jb.map_index[i] : -1;
        mi[w] = (gi >= 0 && gi < jb.map_n) ? gi : -1;
        ++w;
    }
    // scale through the features both maps hold
    const double* Rs = jb.Rt_source;          // [R|t] row-major 3 x 4
    std::vector<size_t> com;
    std::vector<double> ratio;
    for (size_t k = 0; k < nf; ++k) {
        if (mi[k] < 0) continue;
        const double* Xg = jb.map_X + 3 * (size_t)mi[k];
        double xs[3];
        for (int r = 0; r < 3; ++r) xs[r] = Rs[4 * r] * Xg[0] + Rs[4 * r + 1] * Xg[1] + Rs[4 * r + 2] * Xg[2] + Rs[4 * r + 3];
        const double ng = std::sqrt(xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2]);
        const double nt = std::sqrt(Xt[3 * k] * Xt[3 * k] + Xt[3 * k + 1] * Xt[3 * k + 1] + Xt[3 * k + 2] * Xt[3 * k + 2]);
        com.push_back(k);
        ratio.push_back(ng / (nt > 1e-12 ? nt : 1e-12));
    }
    jb.n_common = (int)com.size();
    if (com.size() < 8) return CLC_INTER_NO_SCALE;
    std::vector<double> srt(ratio);
    std::sort(srt.begin(), srt.end());
    const double med = (srt.size() & 1) ? srt[srt.size() / 2] : 0.5 * (srt[srt.size() / 2 - 1] + srt[srt.size() / 2]);
    std::vector<size_t> keep;
    for (size_t k = 0; k < com.size(); ++k) if (std::fabs(ratio[k] / med - 1.0) < 0.2) keep.push_back(com[k]);
    jb.n_common = (int)keep.size();
    if (keep.size() < 8) return CLC_INTER_NO_SCALE;
    double sum = 0.0; size_t good = 0;
    for (size_t k = 0; k + 1 < keep.size(); ++k) {
        const double* g0 = jb.map_X + 3 * (size_t)mi[keep[k]]; const double* g1 = jb.map_X + 3 * (size_t)mi[keep[k + 1]];
        const double* t0 = &Xt[3 * keep[k]]; const double* t1 = &Xt[3 * keep[k + 1]];
        const double d1 = std::sqrt((g1[0] - g0[0]) * (g1[0] - g0[0]) + (g1[1] - g0[1]) * (g1[1] - g0[1]) + (g1[2] - g0[2]) * (g1[2] - g0[2]));
        const double d2 = std::sqrt((t1[0] - t0[0]) * (t1[0] - t0[0]) + (t1[1] - t0[1]) * (t1[1] - t0[1]) + (t1[2] - t0[2]) * (t1[2] - t0[2]));
        if (d2 > 1e-9) { sum += d1 / d2; ++good; }
    }
    if (good == 0) return CLC_INTER_NO_SCALE;
    const double scale = sum / (double)good;
    if (!(scale > 0.0) || !std::isfinite(scale)) return CLC_INTER_NO_SCALE;
    jb.scale = scale;
    // the destination's pose through the source: X_d = R_rel X_s + s t_rel, X_s = R_s X_w + t_s
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 3; ++q) jb.Rt[4 * r + q] = Rb[3 * r] * Rs[q] + Rb[3 * r + 1] * Rs[4 + q] + Rb[3 * r + 2] * Rs[8 + q];
        jb.Rt[4 * r + 3] = Rb[3 * r] * Rs[3] + Rb[3 * r + 1] * Rs[7] + Rb[3 * r + 2] * Rs[11] + scale * tb[r];
    }
    // the temporary map in world coordinates: X_w = R_s^T (s X_tmp - t_s)
    Xw.resize(3 * nf);
    for (size_t k = 0; k < nf; ++k) {
        const double v[3] = { scale * Xt[3 * k] - Rs[3], scale * Xt[3 * k + 1] - Rs[7], scale * Xt[3 * k + 2] - Rs[11] };
        for (int q = 0; q < 3; ++q) Xw[3 * k + q] = Rs[q] * v[0] + Rs[4 + q] * v[1] + Rs[8 + q] * v[2];
    }
    return CLC_INTER_OK;
}

// one refinement staged and enqueued on the context's stream, its record going to pinned memory; returns the `ready` word to poll
int refine_enqueue(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, const double* h_Rt_in, double huber_a,
                   int32_t** ready, double** h_rec)
{
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_d = (size_t)5 * N + 16 + 12;
    const size_t out_d = (pnp_refine_out_bytes() + 7) / 8;
    int rc = ensure_pnp(ctx, in_d + out_d + 8);
    if (rc != CLC_OK) return rc;
    rc = ensure_pinned(ctx, (in_d + out_d) * sizeof(double) + 64);
    if (rc != CLC_OK) return rc;
    double* dX = ctx->d_pnp;
    double* dx = dX + (size_t)3 * N;
    double* dK = dx + (size_t)2 * N;
    double* dRt = dK + 16;
    double* hp = (double*)ctx->h_pin;
    memcpy(hp, h_X, sizeof(double) * 3 * N);
    memcpy(hp + (size_t)3 * N, h_x, sizeof(double) * 2 * N);
    memset(hp + (size_t)5 * N, 0, sizeof(double) * 16);
    memcpy(hp + (size_t)5 * N, h_K, sizeof(double) * 9);
    memcpy(hp + (size_t)5 * N + 16, h_Rt_in, sizeof(double) * 12);
    *h_rec = hp + in_d;
    *ready = (int32_t*)((uint8_t*)*h_rec + pnp_refine_ready_offset());
    __atomic_store_n(*ready, 0, __ATOMIC_RELAXED);
    CLC_HIP(ctx, launch_acr_stage(hp, ctx->d_pnp, (in_d + 1) & ~(size_t)1, ctx->stream));          // inputs by a launch, not a copy command
    CLC_HIP(ctx, launch_pnp_refine(dRt, dX, dx, nullptr, N, dK, huber_a > 0.0 ? huber_a : 16.0, 50, ctx->d_pnp + in_d, ctx->stream, &ctx->prof, nullptr,
                                   *h_rec));
    return CLC_OK;
}

} // namespace

extern "C" {

int clc_essential_acransac_batch(clc_ctx* const* ctxs, clc_two_view_job* jobs, int n_jobs)
{
    if (n_jobs < 0 || (n_jobs > 0 && (!ctxs || !jobs))) return CLC_ERR_BAD_ARG;
    if (n_jobs == 0) return CLC_OK;
    const int rc0 = check_batch_contexts(ctxs, n_jobs, "essential_acransac_batch: every job needs a context of its own");
    if (rc0 != CLC_OK) return rc0;
    std::vector<AcrRun> runs((size_t)n_jobs);
    std::vector<double> EF((size_t)18 * n_jobs, 0.0);
    const bool lockstep = acr_lockstep(1, n_jobs);
    for (int i = 0; i < n_jobs; ++i) two_view_begin(runs[(size_t)i], ctxs[i], jobs[i], &EF[(size_t)18 * i], lockstep, ctxs[0]->stream);
    if (lockstep) drive_group(runs); else drive_runs(runs);
    int worst = CLC_OK;
    for (int i = 0; i < n_jobs; ++i) {
        AcrRun& r = runs[(size_t)i];
        r.finish();
        jobs[i].status = r.status;
        if (jobs[i].E) memcpy(jobs[i].E, &EF[(size_t)18 * i], sizeof(double) * 9);
        if (jobs[i].F) memcpy(jobs[i].F, &EF[(size_t)18 * i + 9], sizeof(double) * 9);
        if (r.status != CLC_OK && worst == CLC_OK) worst = r.status;
    }
    return worst;
}

int clc_inter_pose_batch(clc_ctx* const* ctxs, clc_inter_pose_job* jobs, int n_jobs)
{
    if (n_jobs < 0 || (n_jobs > 0 && (!ctxs || !jobs))) return CLC_ERR_BAD_ARG;
    if (n_jobs == 0) return CLC_OK;
    const int rc0 = check_batch_contexts(ctxs, n_jobs, "inter_pose_batch: every job needs a context of its own");
    if (rc0 != CLC_OK) return rc0;
    for (int i = 0; i < n_jobs; ++i) {
        clc_inter_pose_job& jb = jobs[i];
        if (!jb.tv.E || !jb.tv.inliers || !jb.Rt_source || (jb.map_index && !jb.map_X))
            return fail(ctxs[i], CLC_ERR_BAD_ARG, "inter_pose_batch: a job needs tv.E, tv.inliers, Rt_source (and map_X with map_index)");
        memset(jb.Rt, 0, sizeof jb.Rt); memset(jb.cov, 0, sizeof jb.cov);
        jb.rmse = 0.0; jb.scale = 0.0; jb.n_front = 0; jb.n_common = 0; jb.n_refined = 0; jb.stage = CLC_INTER_NO_MODEL;
    }
    // 1. the a-contrario five-point filters of all pairs, their chains of launches interleaved
    std::vector<AcrRun> runs((size_t)n_jobs);
    std::vector<double> EF((size_t)18 * n_jobs, 0.0);
    const bool lockstep = acr_lockstep(1, n_jobs);
    for (int i = 0; i < n_jobs; ++i) two_view_begin(runs[(size_t)i], ctxs[i], jobs[i].tv, &EF[(size_t)18 * i], lockstep, ctxs[0]->stream);
    if (lockstep) drive_group(runs); else drive_runs(runs);
    int worst = CLC_OK;
    struct Pending { int job; int32_t* ready; double* rec; };
    std::vector<Pending> pend;
    std::vector<std::vector<double>> Xw((size_t)n_jobs), x2f((size_t)n_jobs);
    for (int i = 0; i < n_jobs; ++i) {
        AcrRun& r = runs[(size_t)i];
        clc_inter_pose_job& jb = jobs[i];
        r.finish();
        jb.tv.status = r.status;
        memcpy(jb.tv.E, &EF[(size_t)18 * i], sizeof(double) * 9);
        if (jb.tv.F) memcpy(jb.tv.F, &EF[(size_t)18 * i + 9], sizeof(double) * 9);
        if (r.status != CLC_OK) { if (worst == CLC_OK) worst = r.status; continue; }
        // 2. host geometry, 3. the refinement enqueued on the job's own context
        jb.stage = inter_geometry(jb, Xw[(size_t)i], x2f[(size_t)i]);
        if (jb.stage != CLC_INTER_OK) continue;
        Pending p{ i, nullptr, nullptr };
        const int rc = refine_enqueue(ctxs[i], Xw[(size_t)i].data(), x2f[(size_t)i].data(), jb.n_front, jb.tv.K2, jb.Rt, jb.huber_a, &p.ready, &p.rec);
        if (rc != CLC_OK) { jb.tv.status = rc; jb.stage = CLC_INTER_NO_REFINEMENT; if (worst == CLC_OK) worst = rc; continue; }
        pend.push_back(p);
    }
    // 4. collect: poll the pinned records, fall back to the stream synchronisation after 5 ms
    const auto t0 = std::chrono::steady_clock::now();
    for (const Pending& p : pend) {
        clc_inter_pose_job& jb = jobs[p.job];
        uint32_t spins = 0;
        while (__atomic_load_n(p.ready, __ATOMIC_ACQUIRE) == 0) {
            if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
                const hipError_t e = hipStreamSynchronize(ctxs[p.job]->stream);
                if (e != hipSuccess || __atomic_load_n(p.ready, __ATOMIC_ACQUIRE) == 0) {
                    jb.tv.status = fail(ctxs[p.job], CLC_ERR_HIP, "inter_pose_batch: refinement did not complete", e);
                    jb.stage = CLC_INTER_NO_REFINEMENT;
                    if (worst == CLC_OK) worst = jb.tv.status;
                    break;
                }
            }
        }
        if (jb.stage != CLC_INTER_OK) continue;
        struct { double Rt[12]; double cov[36]; double cost; double rmse; int32_t iterations; int32_t n_used; } f;
        memcpy(&f, p.rec, sizeof f);
        memcpy(jb.Rt, f.Rt, sizeof f.Rt);
        memcpy(jb.cov, f.cov, sizeof f.cov);
        jb.rmse = f.rmse;
        jb.n_refined = f.n_used;
    }
    return worst;
}

} // extern "C"

extern "C" {

int clc_cov_intersection(const double* CA, const double* CB, const double* ca, const double* cb, double* omega,
                         double* cov_fused, double* pos_fused)
{
    if (!CA || !CB || !ca || !cb) return CLC_ERR_BAD_ARG;
    coloc::Mat3d A, B;
    coloc::Vec3d a, b;
    for (int i = 0; i < 9; ++i) { A[i] = CA[i]; B[i] = CB[i]; }
    for (int i = 0; i < 3; ++i) { a[i] = ca[i]; b[i] = cb[i]; }
    coloc::HIPCovIntersection ci;
    ci.loadData(A, B, a, b);
    ci.optimize();
    ci.computeFusedValues();
    if (omega) *omega = ci.minX;
    if (cov_fused) for (int i = 0; i < 9; ++i) cov_fused[i] = ci.covFused[i];
    if (pos_fused) for (int i = 0; i < 3; ++i) pos_fused[i] = ci.poseFused[i];
    return CLC_OK;
}

} // extern "C"
