// clc_ctx.h -- the context behind the C ABI and the helpers the capi_*.hip translation units share (not installed).
#ifndef CLC_CTX_H
#define CLC_CTX_H

#include "clc_internal.h"

#include <string>
#include <vector>

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
} // namespace clc

namespace clc { struct DescEntry; }

struct clc_ctx {
    int device = 0;
    std::vector<float> acr_lg;   // (float) log10(k), k = 0 .. : the a-contrario tables are sums over it
    hipStream_t stream = nullptr;
    std::string err;
    bool has_det = false, has_mat = false;
    clc_detector_opts dopts{};
    clc_matcher_opts mopts{};
    // pyramid
    clc::PyramidDesc pd{};
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
    int formulation = clc::K2NN_MATRIX;   // K2NN sweep formulation (k2nn.hip): FP4 matrix pipe, or round 1's popcount kernel for A/B runs
    int target_blocks = 0;           // K2NN sweep workgroups aimed at per launch; 0 = the formulation's default
    int bias_a = 326, bias_b = 249;  // matrix sweep, one-round single-job plans: train share of a workgroup on wave slot 0 / 1 in 1/256 of the
                                     // equal share (k2nn.hip; measured optimum 21 : 16 : 12-13 tiles at 10k x 10k); CLC_K2NN_BIAS=a,b, 0,0 = equal shares
    bool xcd_map = true;         // XCD-aware K2NN tile order (CLC_K2NN_XCD_MAP=0 switches it off for A/B runs)
    clc::K2nnDevice k2dev{};          // XCDs and CUs of this context's device (the sweep planner's balance arguments)
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
    // host front end (clc_detect_and_describe*): ONE pinned block [ image | keypoints | descriptors | {written, found} ] the frame goes
    // in and out through, and the block of the descriptor table (desc_cache.h) the frame's descriptors are written into on the device
    uint8_t* h_stage = nullptr;
    size_t stage_img = 0, stage_kps = 0, stage_desc = 0, stage_cnt = 0;      // byte offsets inside h_stage
    clc::DescEntry* desc_pending = nullptr;
    int staged_n = -1;            // rows of the last staged frame (-1: none)
    // host-pointer match entries: pinned mirror of the results (so that the host can verify published blocks while the GPU sweeps)
    uint8_t* h_res = nullptr;
    size_t res_cap = 0;
    clc::Profiler prof;
};

namespace clc {

// records the failure text on the context (clc_last_error_string) and returns `code`
int fail(clc_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess);

#define CLC_HIP(ctx, call)                                                      \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) return ::clc::fail((ctx), CLC_ERR_HIP, #call, e__);     \
    } while (0)

inline hipStream_t pick(clc_ctx* ctx, void* stream) { return stream ? (hipStream_t)stream : ctx->stream; }

// context-owned workspaces, grown on demand (capi_core.hip)
int ensure_partial(clc_ctx* ctx, size_t elems);           // armed K2NN top-2 rows + arrival counters
int ensure_pnp(clc_ctx* ctx, size_t doubles);             // pose scratch
int ensure_pinned(clc_ctx* ctx, size_t bytes);            // pinned staging of the pose solves
int ensure_slots(clc_ctx* ctx, int n, hipStream_t st);    // pyramids (+ detector maps) of n cameras
int ensure_results(clc_ctx* ctx, size_t bytes);           // pinned mirror of match results

// K2NN: plan + launch a job list on `st` with the context's formulation / shares (capi_match.hip)
int default_target_blocks(int formulation, const K2nnDevice& dev = K2nnDevice{});
int run_jobs(clc_ctx* ctx, std::vector<K2nnJobDev>& jobs, hipStream_t st, bool probe = false);
void k2nn_probe_bias(clc_ctx* ctx);                       // once per process and device: which unequal shares suit this device

// all two-view a-contrario filters of a batch (pose_batch.hip); jobs[i] on ctxs[i]
int acr_two_view_batch(clc_ctx* const* ctxs, clc_two_view_job* const* jobs, int n_jobs, int kind /* 1 'E', 2 'F', 3 'H' */);
int check_batch_contexts(clc_ctx* const* ctxs, int n_jobs, const char* what);

} // namespace clc
#endif
