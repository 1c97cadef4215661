// capi_core.hip -- the C ABI of libcoloc_hip.so (include/coloc_hip.h): context, status codes, kernel timing, and the front end
// (pyramid, detector, CLATCH entry points).
//
// Host-side replacement for the CUDA-runtime plumbing of the reference's include/coloc/GPUDetector.hpp (ctor :70-138,
// detectAndDescribe :216-291).  No textures, no per-call object creation (the reference leaks a texture object per frame,
// GPUDetector.hpp:240,244), one stream per context, every HIP status checked and mapped to an int code.
// The matcher entry points are in capi_match.hip, the descriptor hand-over in desc_cache.hip, the pose solvers in capi_pose.hip /
// pose_batch.hip / inter_pose.hip.
#include "clc_ctx.h"
#include "desc_cache.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace clc;

namespace clc {
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

int fail(clc_ctx* ctx, int code, const char* what, hipError_t e)
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

// Host staging for the robust pose solve: ONE pinned buffer, ONE H2D copy in, ONE D2H copy out.
int ensure_pinned(clc_ctx* ctx, size_t bytes)
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

// room for the pyramids (+ score maps, keypoint masks, tile counts) of n cameras; slot 0 (the current single-image pyramid) does not
// survive a growth -- every caller rebuilds it
int ensure_slots(clc_ctx* ctx, int n, hipStream_t st)
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


int ensure_results(clc_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->res_cap) return CLC_OK;
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_res) CLC_HIP(ctx, hipHostFree(ctx->h_res));
    ctx->h_res = nullptr; ctx->res_cap = 0;
    CLC_HIP(ctx, hipHostMalloc((void**)&ctx->h_res, bytes, hipHostMallocDefault));
    ctx->res_cap = bytes;
    return CLC_OK;
}

} // namespace clc

namespace {

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

int cache_mode_default()
{
    const char* v = getenv("CLC_DESC_CACHE");
    if (!v || !v[0]) return CLC_DESC_CACHE_VERIFY;
    if (v[0] == '0' || v[0] == 'o' || v[0] == 'O') return CLC_DESC_CACHE_OFF;
    if (v[0] == 't' || v[0] == 'T' || v[0] == '2') return CLC_DESC_CACHE_TRUST;
    return CLC_DESC_CACHE_VERIFY;
}


// The host front end's way out (clc_detect_and_describe_view): the last launch of a frame mirrors the count, the keypoints and the
// descriptors that were found -- nothing past the count -- into the context's pinned block; the frame's one stream synchronisation
// follows.  Round 6 built and measured two other forms on the same frames (640 x 480, ~3 800 keypoints; profiles/r06_policy_path_notes.txt):
// (a) no synchronisation at all -- keypoints mirrored behind the detector by a launch whose last workgroup publishes a polled word behind
// system-scope fences, the CLATCH waves storing their descriptors twice, a one-thread launch marking them complete: 114 us per frame
// against 94-97 (the fences 18 us, CLATCH + 5 us, 4 us for a 4-byte store to be acknowledged over PCIe); (b) two mirror launches and an
// event behind the first, so that the host converts keypoints under CLATCH: 96-98 us (two more runtime calls in the enqueue phase and
// 5 us of keypoint mirror in front of CLATCH take back what the overlap gives).
__global__ __launch_bounds__(256) void frontend_mirror_kernel(const uint32_t* __restrict__ d_kps, const uint4* __restrict__ d_desc,
                                                              const uint32_t* __restrict__ d_count, uint32_t* __restrict__ h_kps,
                                                              uint4* __restrict__ h_desc, uint32_t* __restrict__ h_count)
{
    const uint32_t n = d_count[0];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0u) { h_count[0] = n; h_count[1] = d_count[1]; }
    if (i < n * 4u) h_desc[i] = d_desc[i];                       // 64 B per row = four uint4
    // 20 B per keypoint = five dwords: the first 4 n by the same threads, the rest by the first n of them
    if (i < n * 4u) h_kps[i] = d_kps[i];
    if (i < n) h_kps[n * 4u + i] = d_kps[n * 4u + i];
}

static int ensure_stage(clc_ctx* ctx)
{
    if (ctx->h_stage) return CLC_OK;
    const size_t img = ((size_t)ctx->dopts.width * ctx->dopts.height + 255) & ~(size_t)255;
    const size_t kps = ((size_t)ctx->dopts.maxkp * sizeof(clc_keypoint) + 255) & ~(size_t)255;
    const size_t desc = (size_t)ctx->dopts.maxkp * CLC_DESC_BYTES;
    ctx->stage_img = 0; ctx->stage_kps = img; ctx->stage_desc = img + kps; ctx->stage_cnt = img + kps + desc;
    CLC_HIP(ctx, hipHostMalloc((void**)&ctx->h_stage, img + kps + desc + 256, hipHostMallocDefault));
    memset(ctx->h_stage + ctx->stage_cnt, 0, 256);          // {written, found}
    return CLC_OK;
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
    desc_drop_owner(ctx);                  // freeGPUMemory: what this context published dies with it
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_res) (void)hipHostFree(ctx->h_res);
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

/* ---- the whole front end on host buffers: GPUDetector::detectAndDescribe (GPUDetector.hpp:216-291) as one enqueue sequence ----------
 * The frame goes in and out through ONE pinned block owned by the context: the image is copied into it and uploaded by the DMA engine,
 * pyramid / detector / CLATCH run back to back, and a last small launch mirrors the count, the keypoints and the descriptors that were
 * found -- nothing past the count -- into the pinned block.  One stream synchronisation per frame (the reference: 7 level downloads
 * with a synchronisation each, KFAST on the host, two uploads, a download and a device synchronisation, :262-290).  The descriptors are
 * written on the device into a block of the descriptor table (desc_cache.h), so that publishing them needs no device copy. */
int clc_detect_and_describe_view(clc_ctx* ctx, const uint8_t* h_img, uint32_t width, uint32_t height, const clc_keypoint** h_kps,
                                 const uint8_t** h_desc, int* n_written, int* n_found)
{
    if (h_kps) *h_kps = nullptr;
    if (h_desc) *h_desc = nullptr;
    if (n_written) *n_written = 0;
    if (n_found) *n_found = 0;
    if (!ctx || !h_img) return fail(ctx, CLC_ERR_BAD_ARG, "detect_and_describe: bad argument");
    if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "detect_and_describe: context created without detector options");
    if (width != ctx->dopts.width || height != ctx->dopts.height)
        return fail(ctx, CLC_ERR_BAD_ARG, "detect_and_describe: image size differs from DetectorOptions width/height");
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    { const int rc = ensure_stage(ctx); if (rc != CLC_OK) return rc; }
    ctx->staged_n = -1;
    // where the descriptors go on the device: a block of the descriptor table that nothing published stands on (a published block
    // stays as it is until its publication dies), else the context's own array
    uint8_t* d_rows = nullptr;
    if (ctx->cache_mode != CLC_DESC_CACHE_OFF) {
        if (!ctx->desc_pending) ctx->desc_pending = desc_reserve(ctx, ctx->device, ctx->dopts.maxkp, &d_rows);
        else d_rows = desc_rows(ctx->desc_pending);
    }
    if (!d_rows) d_rows = (uint8_t*)ctx->d_desc;
    const LevelDesc& L0 = ctx->pd.lv[0];
    uint8_t* hp = ctx->h_stage;
    uint32_t* h_cnt = (uint32_t*)(hp + ctx->stage_cnt);
    memcpy(hp + ctx->stage_img, h_img, (size_t)width * height);
    // (the DMA engine; a compute-queue copy kernel reading the pinned block was measured 3 us slower per frame, round 6)
    if (L0.pitch == width)
        CLC_HIP(ctx, hipMemcpyAsync(ctx->d_arena + L0.offset, hp + ctx->stage_img, (size_t)width * height, hipMemcpyHostToDevice, ctx->stream));
    else
        CLC_HIP(ctx, hipMemcpy2DAsync(ctx->d_arena + L0.offset, L0.pitch, hp + ctx->stage_img, width, width, height, hipMemcpyHostToDevice, ctx->stream));
    CLC_HIP(ctx, launch_pyramid(ctx->pd, ctx->d_arena, ctx->d_arena + L0.offset, L0.pitch, ctx->stream, &ctx->prof));
    ctx->pyramid_valid = true;
    int rc = clc_detect_dev(ctx, nullptr);
    if (rc != CLC_OK) return rc;
    rc = clc_describe_detected_dev(ctx, d_rows, nullptr);
    if (rc != CLC_OK) return rc;
    const uint32_t blocks = (ctx->dopts.maxkp * 4u + 255u) / 256u;
    hipLaunchKernelGGL(frontend_mirror_kernel, dim3(blocks ? blocks : 1u), dim3(256), 0, ctx->stream, (const uint32_t*)ctx->d_kps, (const uint4*)d_rows,
                       (const uint32_t*)ctx->d_count, (uint32_t*)(hp + ctx->stage_kps), (uint4*)(hp + ctx->stage_desc), h_cnt);
    CLC_HIP(ctx, hipGetLastError());
    CLC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int n = (int)h_cnt[0];
    ctx->staged_n = n;
    if (h_kps) *h_kps = (const clc_keypoint*)(hp + ctx->stage_kps);
    if (h_desc) *h_desc = hp + ctx->stage_desc;
    if (n_written) *n_written = n;
    if (n_found) *n_found = (int)h_cnt[1];
    return CLC_OK;
}

int clc_detect_and_describe(clc_ctx* ctx, const uint8_t* h_img, uint32_t width, uint32_t height, clc_keypoint* h_kps,
                            uint8_t* h_desc, int capacity, int* n_written, int* n_found)
{
    if (capacity < 0 || (capacity > 0 && (!h_kps || !h_desc))) return fail(ctx, CLC_ERR_BAD_ARG, "detect_and_describe: bad argument");
    const clc_keypoint* kp = nullptr;
    const uint8_t* ds = nullptr;
    int n = 0;
    const int rc = clc_detect_and_describe_view(ctx, h_img, width, height, &kp, &ds, &n, n_found);
    if (rc != CLC_OK) return rc;
    if (n > capacity) n = capacity;
    if (n > 0) {
        memcpy(h_kps, kp, (size_t)n * sizeof(clc_keypoint));
        memcpy(h_desc, ds, (size_t)n * CLC_DESC_BYTES);
    }
    if (n_written) *n_written = n;
    return CLC_OK;
}

int clc_detect_store_descriptors(clc_ctx* ctx, void* h_dst, int n, clc_desc_handle* handle)
{
    if (handle) memset(handle, 0, sizeof *handle);
    if (!ctx || n < 0 || (n > 0 && !h_dst)) return fail(ctx, CLC_ERR_BAD_ARG, "detect_store_descriptors: bad argument");
    if (!ctx->h_stage || ctx->staged_n < 0 || n > ctx->staged_n) return fail(ctx, CLC_ERR_STATE, "detect_store_descriptors: no staged frame of that many rows");
    if (n == 0) return CLC_OK;
    const uint8_t* src = ctx->h_stage + ctx->stage_desc;
    // a partial store (fewer rows than were found) or a context that does not publish: the copy alone
    if (ctx->cache_mode == CLC_DESC_CACHE_OFF || !ctx->desc_pending || n != ctx->staged_n) {
        memcpy(h_dst, src, (size_t)n * CLC_DESC_BYTES);
        return CLC_OK;
    }
    // the single copy of the frame's descriptors into the caller's block, folded on the way; the device rows are in the table already
    const uint64_t fold = desc_copy_fold(h_dst, src, (size_t)n);
    desc_publish(ctx->desc_pending, h_dst, n, fold, true, handle);
    ctx->desc_pending = nullptr;                                  // the next frame takes another block
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

} // extern "C"
