// multicam.hip -- the multi-camera step behind the C ABI: one camera per GPU (one process per GPU), exchange of the
// 512-bit descriptors over xGMI, then every rank sweeps its share of the all-pairs matching.
//
// The reference has no multi-GPU path: the all-pairs loop is serial over Utils::handlePairs(n) = exhaustivePairs in
// one process (include/coloc/GPUMatcher.hpp:143-155, colocUtils.hpp:58-61).  This keeps that loop's RESULT (per
// (first < second) pair: Q = descriptors of `first`, T = of `second`, GPUMatcher.hpp:165-172) and re-cuts the work
// (SURVEY.md 8e): after the exchange every rank holds every camera's descriptors, so the flattened
// (pair, query-block) sequence is dealt out in equal contiguous shares -- no second data-path collective.
//
// Two exchange forms, selectable per call (the payload is 640 KB per rank at 10k keypoints: latency-bound, every peer
// block crosses exactly one xGMI link of the full mesh):
//   CLC_MC_RCCL      ncclAllGather of the fixed-capacity block (+ a 4-byte all-gather of the counts);
//   CLC_MC_PEER_COPY one-shot fan-out: every rank writes its block straight into each peer's arena through IPC-mapped
//                    pointers (hipMemcpyAsync device-to-device, one xGMI hop each), and the counts' all-gather that
//                    follows on the same stream is the fence -- a rank contributes only after its copies have drained.
// Ordering contract of the peer copies (round 3): the arena is DOUBLE-BUFFERED by step parity.  Step k's blocks land in
// buffer k & 1 while a peer may still be sweeping step k - 1 out of buffer (k - 1) & 1; buffer k & 1 is written again by
// step k + 2, whose copies a rank enqueues behind its own part of step k + 1's count collective -- and that collective
// completes only when EVERY rank has joined it, i.e. has its step-k sweep (enqueued earlier on the same stream) behind
// it.  So the skew between ranks is bounded to one step and no copy ever lands in a buffer a sweep still reads.  Every
// rank must call the gather the same number of times (the parity is a local call count).
// OVERLAPPED steps (round 6, clc_mc_set_overlap; default off): step k + 1's exchange -- and the describe the caller enqueues in front of
// it -- runs on ONE stream while step k's sweep runs on ANOTHER.  All collectives stay on the exchange stream (one communicator is
// enough: a rank's collectives keep their order), the sweep has none.  The arena then has THREE buffers, buffer j % 3 for step j, and
// two event chains carry the ordering the single stream gave for free:
//   sweep j waits for exchange j (event behind the exchange);
//   exchange j waits for sweep j - 2 (event behind that sweep) before it enqueues anything.
// Why that is enough, peer copies included: buffer j % 3 was last read by sweep j - 3, which precedes sweep j - 2 on the sweep stream.
// A peer's copy of step j into this rank's buffer is enqueued behind the peer's count collective of step j - 1, which completes only
// when THIS rank has joined it -- on its exchange stream, behind its wait for sweep (j - 1) - 2 = j - 3.  With two buffers the same
// argument would need exchange j to wait for sweep j - 1, i.e. no overlap.
// RCCL is resolved at run time (dlopen "librccl.so.1": the copy the process already has, e.g. PyTorch's, or the
// system one) and its handful of types is declared here, so libcoloc_hip.so needs neither the RCCL headers to build
// nor the library to load, and single-GPU hosts never touch it.
//
// Written against the public C ABI + the HIP runtime only.  The planner (clc_mc_plan) is pure host arithmetic and is
// the C twin of coloc_amd/multicam.py shard_pairs (tests/test_multicam.py checks them against each other).
#include "clc_internal.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

extern "C" int clc_ctx_device(const clc_ctx* ctx);

namespace {

// the part of the (stable) NCCL / RCCL C API this file uses, declared locally: nccl.h's ncclUniqueId is 128 opaque bytes, a
// communicator is an opaque pointer, results and data types are plain enums (ncclSuccess == 0, ncclUint8 == 1)
struct ncclUniqueId { char internal[CLC_MC_ID_BYTES]; };
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclUint8 = 1;

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    if (r.lib) return r;
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) return r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
    return r;
}

// A count travels to the device BY VALUE, as the argument of a one-thread launch.  (Round 3 staged it in one pinned word per buffer and
// copied that asynchronously: the copy reads its source when it EXECUTES, so a host running two steps ahead -- the enqueue-only gather
// never synchronises -- rewrote the word of the same parity before the earlier copy had run.)
__global__ void mc_set_count_kernel(int32_t* __restrict__ dst, const int32_t value) { *dst = value; }

} // namespace

struct clc_mc {
    clc_ctx* ctx = nullptr;
    int world = 1, rank = 0, cap = 0, device = 0;
    ncclComm_t comm = nullptr;
    int nbuf = 2;                          // arena buffers: 2 (steps on one stream), 3 (overlapped steps)
    uint8_t* d_arena = nullptr;            // [nbuf][world][cap][64]: buffer (step % nbuf) receives step's blocks
    int32_t* d_counts = nullptr;           // [nbuf][world + 1]: the gathered counts of a buffer, this rank's own count at [world]
    int32_t* h_counts = nullptr;           // pinned, [nbuf][world + 1]: host mirror of the gathered counts (written by the device only)
    bool overlap = false;                  // exchange and sweep on different streams (clc_mc_set_overlap)
    hipEvent_t ev_exch[3] = {}, ev_sweep[3] = {};   // behind the exchange / the sweep of the step a buffer holds
    long exch_step[3] = { -1, -1, -1 }, sweep_step[3] = { -1, -1, -1 };   // which step those events stand for (-1: none recorded)
    long steps = 0;                        // exchanges enqueued so far = the number of the next step
    int fill = 0;                          // buffer the NEXT gather (and clc_mc_virtual_put) writes
    int cur = 0;                           // buffer of the last completed gather: what the match entries sweep
    bool counts_on_host = false;           // the last gather synchronised and left the counts in h_counts[cur]
    std::vector<uint8_t*> peer_arena;      // IPC-mapped arenas of the peers (own pointer at [rank]); empty until first used
    bool peers_tried = false;
    bool is_virtual = false;
    std::string err;
};

namespace {

int mc_fail(clc_mc* mc, int code, const char* what, hipError_t e = hipSuccess, ncclResult_t n = ncclSuccess)
{
    if (mc) {
        char buf[384];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else if (n != ncclSuccess) snprintf(buf, sizeof buf, "%s: rccl error %d (%s)", what, (int)n, rccl().GetErrorString ? rccl().GetErrorString(n) : "?");
        else snprintf(buf, sizeof buf, "%s", what);
        mc->err = buf;
    }
    return code;
}
#define MC_HIP(mc, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return mc_fail((mc), CLC_ERR_HIP, #call, e__); } while (0)
#define MC_NCCL(mc, call) do { ncclResult_t n__ = (call); if (n__ != ncclSuccess) return mc_fail((mc), CLC_ERR_HIP, #call, hipSuccess, n__); } while (0)

// the arena and the count rows for mc->nbuf buffers (again after clc_mc_set_overlap)
int mc_alloc(clc_mc* mc)
{
    if (mc->d_arena) (void)hipFree(mc->d_arena);
    if (mc->d_counts) (void)hipFree(mc->d_counts);
    if (mc->h_counts) (void)hipHostFree(mc->h_counts);
    mc->d_arena = nullptr; mc->d_counts = nullptr; mc->h_counts = nullptr;
    const size_t nb = (size_t)mc->nbuf, rows = sizeof(int32_t) * (size_t)(mc->world + 1);
    MC_HIP(mc, hipMalloc((void**)&mc->d_arena, nb * (size_t)mc->world * (size_t)mc->cap * CLC_DESC_BYTES));
    MC_HIP(mc, hipMalloc((void**)&mc->d_counts, nb * rows));
    MC_HIP(mc, hipMemset(mc->d_counts, 0, nb * rows));
    MC_HIP(mc, hipHostMalloc((void**)&mc->h_counts, nb * rows, hipHostMallocDefault));
    memset(mc->h_counts, 0, nb * rows);
    mc->fill = 0; mc->cur = 0;
    return CLC_OK;
}

// exchange the arenas' IPC handles once (through the communicator itself) and map the peers' arenas
int open_peers(clc_mc* mc, hipStream_t st)
{
    if (!mc->peer_arena.empty()) return CLC_OK;
    if (mc->peers_tried) return mc_fail(mc, CLC_ERR_STATE, "peer copy: mapping the peers' arenas failed earlier");
    mc->peers_tried = true;
    std::vector<uint8_t*> peers((size_t)mc->world, nullptr);
    peers[(size_t)mc->rank] = mc->d_arena;
    if (mc->world > 1) {
        hipIpcMemHandle_t mine;
        MC_HIP(mc, hipIpcGetMemHandle(&mine, mc->d_arena));
        hipIpcMemHandle_t* d_h = nullptr;
        MC_HIP(mc, hipMalloc((void**)&d_h, sizeof(hipIpcMemHandle_t) * (size_t)(mc->world + 1)));
        std::vector<hipIpcMemHandle_t> all((size_t)mc->world);
        hipError_t e = hipMemcpyAsync(d_h + mc->world, &mine, sizeof mine, hipMemcpyHostToDevice, st);
        ncclResult_t n = ncclSuccess;
        if (e == hipSuccess) n = rccl().AllGather(d_h + mc->world, d_h, sizeof(hipIpcMemHandle_t), ncclUint8, mc->comm, st);
        if (e == hipSuccess && n == ncclSuccess) e = hipMemcpyAsync(all.data(), d_h, sizeof(hipIpcMemHandle_t) * (size_t)mc->world, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && n == ncclSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_h);
        if (e != hipSuccess || n != ncclSuccess) return mc_fail(mc, CLC_ERR_HIP, "peer copy: IPC handle exchange", e, n);
        for (int p = 0; p < mc->world; ++p) {
            if (p == mc->rank) continue;
            void* ptr = nullptr;
            MC_HIP(mc, hipIpcOpenMemHandle(&ptr, all[(size_t)p], hipIpcMemLazyEnablePeerAccess));
            peers[(size_t)p] = (uint8_t*)ptr;
        }
    }
    mc->peer_arena.swap(peers);
    return CLC_OK;
}

} // namespace

extern "C" {

int clc_mc_plan(const int* counts, int ncams, int world, int rank, int grain, clc_mc_share* out, int capacity, int* n_out)
{
    if (!counts || ncams < 0 || world < 1 || rank < 0 || rank >= world || grain < 1 || capacity < 0 || (capacity > 0 && !out) || !n_out)
        return CLC_ERR_BAD_ARG;
    // flattened (pair, query block) sequence of the pairs whose two sides are non-empty, pairs in exhaustivePairs order
    struct Blk { int32_t first, second; uint32_t blocks; };
    std::vector<Blk> pairs;
    uint64_t total = 0;
    for (int i = 0; i < ncams; ++i)
        for (int j = i + 1; j < ncams; ++j) {
            if (counts[i] < 0 || counts[j] < 0) return CLC_ERR_BAD_ARG;
            if (counts[i] == 0 || counts[j] == 0) continue;        // an empty side yields no matches (GPUMatcher.hpp:150)
            const uint32_t nb = ((uint32_t)counts[i] + (uint32_t)grain - 1u) / (uint32_t)grain;
            pairs.push_back({ i, j, nb });
            total += nb;
        }
    const uint64_t lo = total * (uint64_t)rank / (uint64_t)world, hi = total * (uint64_t)(rank + 1) / (uint64_t)world;
    int n = 0;
    uint32_t out_off = 0;
    uint64_t base = 0;
    for (const Blk& p : pairs) {
        const uint64_t b0 = lo > base ? lo - base : 0, b1 = hi - base < p.blocks ? hi - base : p.blocks;   // [b0, b1) of this pair
        if (hi > base && b0 < b1) {
            const uint32_t q_begin = (uint32_t)b0 * (uint32_t)grain;
            const uint32_t q_end = (uint32_t)b1 * (uint32_t)grain < (uint32_t)counts[p.first] ? (uint32_t)b1 * (uint32_t)grain : (uint32_t)counts[p.first];
            if (n >= capacity) return CLC_ERR_CAPACITY;
            out[n].first = p.first; out[n].second = p.second;
            out[n].q_begin = q_begin; out[n].nq = q_end - q_begin; out[n].out_offset = out_off;
            out_off += q_end - q_begin;
            ++n;
        }
        base += p.blocks;
        if (base >= hi) break;
    }
    *n_out = n;
    return CLC_OK;
}

int clc_mc_unique_id(uint8_t id[CLC_MC_ID_BYTES])
{
    if (!id) return CLC_ERR_BAD_ARG;
    if (!rccl().ok) return CLC_ERR_STATE;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return CLC_ERR_HIP;
    memcpy(id, u.internal, CLC_MC_ID_BYTES);
    return CLC_OK;
}

int clc_mc_create(clc_ctx* ctx, const uint8_t id[CLC_MC_ID_BYTES], int world, int rank, int maxkp, clc_mc** out)
{
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || maxkp < 1) return CLC_ERR_BAD_ARG;
    *out = nullptr;
    clc_mc* mc = new (std::nothrow) clc_mc;
    if (!mc) return CLC_ERR_HIP;
    mc->ctx = ctx; mc->world = world; mc->rank = rank; mc->cap = maxkp; mc->device = clc_ctx_device(ctx);
    auto bail = [&](int code) { clc_mc_destroy(mc); return code; };
    if (hipSetDevice(mc->device) != hipSuccess) return bail(CLC_ERR_HIP);
    mc->is_virtual = world > 1 && !id;      // rehearsal group: no communicator, the caller plays the other ranks (clc_mc_virtual_put)
    if (id) {                                                       // world == 1 with an id: a one-rank communicator (the same RCCL calls as
                                                                    // world > 1; tests/test_gpu_multicam.py drives the run-time-resolved ABI through it)
        if (!rccl().ok) return bail(CLC_ERR_STATE);                 // no RCCL in this process / on this machine
        ncclUniqueId u;
        memcpy(u.internal, id, CLC_MC_ID_BYTES);
        if (rccl().CommInitRank(&mc->comm, world, u, rank) != ncclSuccess) return bail(CLC_ERR_HIP);
    }
    const int rc = mc_alloc(mc);
    if (rc != CLC_OK) return bail(rc);
    *out = mc;
    return CLC_OK;
}

int clc_mc_set_overlap(clc_mc* mc, int on)
{
    if (!mc) return CLC_ERR_BAD_ARG;
    if (mc->steps != 0 || !mc->peer_arena.empty()) return mc_fail(mc, CLC_ERR_STATE, "mc_set_overlap: only before the first exchange (and before clc_mc_open_peers)");
    if ((on != 0) == mc->overlap) return CLC_OK;
    MC_HIP(mc, hipSetDevice(mc->device));
    MC_HIP(mc, hipDeviceSynchronize());
    mc->overlap = on != 0;
    mc->nbuf = mc->overlap ? 3 : 2;
    if (mc->overlap)
        for (int b = 0; b < 3; ++b) {
            if (!mc->ev_exch[b]) MC_HIP(mc, hipEventCreateWithFlags(&mc->ev_exch[b], hipEventDisableTiming));
            if (!mc->ev_sweep[b]) MC_HIP(mc, hipEventCreateWithFlags(&mc->ev_sweep[b], hipEventDisableTiming));
        }
    return mc_alloc(mc);
}

int clc_mc_comm_info(const clc_mc* mc, int* n_ranks, int* user_rank)
{
    if (!mc) return CLC_ERR_BAD_ARG;
    if (n_ranks) *n_ranks = 0;                       // 0: no communicator behind this handle (one rank without an id, or a rehearsal handle)
    if (user_rank) *user_rank = -1;
    if (!mc->comm) return CLC_OK;
    if (!rccl().CommCount || !rccl().CommUserRank) return CLC_ERR_STATE;
    int n = 0, r = -1;
    if (rccl().CommCount(mc->comm, &n) != ncclSuccess || rccl().CommUserRank(mc->comm, &r) != ncclSuccess) return CLC_ERR_HIP;
    if (n_ranks) *n_ranks = n;
    if (user_rank) *user_rank = r;
    return CLC_OK;
}

int clc_mc_destroy(clc_mc* mc)
{
    if (!mc) return CLC_ERR_BAD_ARG;
    (void)hipSetDevice(mc->device);
    for (int p = 0; p < (int)mc->peer_arena.size(); ++p)
        if (p != mc->rank && mc->peer_arena[(size_t)p]) (void)hipIpcCloseMemHandle(mc->peer_arena[(size_t)p]);
    if (mc->comm && rccl().ok) (void)rccl().CommDestroy(mc->comm);
    for (int b = 0; b < 3; ++b) { if (mc->ev_exch[b]) (void)hipEventDestroy(mc->ev_exch[b]); if (mc->ev_sweep[b]) (void)hipEventDestroy(mc->ev_sweep[b]); }
    if (mc->d_arena) (void)hipFree(mc->d_arena);
    if (mc->d_counts) (void)hipFree(mc->d_counts);
    if (mc->h_counts) (void)hipHostFree(mc->h_counts);
    delete mc;
    return CLC_OK;
}

const char* clc_mc_last_error_string(const clc_mc* mc) { return mc ? mc->err.c_str() : "null handle"; }

int clc_mc_arena(const clc_mc* mc, void** d_arena, int* world, int* maxkp)
{
    if (!mc) return CLC_ERR_BAD_ARG;
    if (d_arena) *d_arena = mc->d_arena + (size_t)mc->cur * (size_t)mc->world * (size_t)mc->cap * CLC_DESC_BYTES;
    if (world) *world = mc->world;
    if (maxkp) *maxkp = mc->cap;
    return CLC_OK;
}

// the exchange itself, enqueue only: this rank's block and count into buffer `fill` of every rank
static int mc_exchange(clc_mc* mc, const void* d_my_desc, int my_count, const int32_t* d_my_count, int mode, hipStream_t st)
{
    const size_t block = (size_t)mc->cap * CLC_DESC_BYTES, buf = (size_t)mc->world * block;
    const int b = mc->fill;
    const long step = mc->steps;
    if (mc->overlap && step >= 2) {
        // nothing of step `step` is enqueued before the sweep of step - 2 has ended (top of this file)
        // (every recorded sweep up to step - 2: a caller that skipped a step's sweep still gets the buffer it is about to fill back)
        for (int bw = 0; bw < 3; ++bw)
            if (mc->sweep_step[bw] >= 0 && mc->sweep_step[bw] <= step - 2) MC_HIP(mc, hipStreamWaitEvent(st, mc->ev_sweep[bw], 0));
    }
    uint8_t* arena = mc->d_arena + (size_t)b * buf;
    int32_t* d_cnt = mc->d_counts + (size_t)b * (size_t)(mc->world + 1);
    int32_t* h_cnt = mc->h_counts + (size_t)b * (size_t)(mc->world + 1);
    // rows to move: the valid ones when the host knows the count, the whole fixed-capacity block when only the device does
    const size_t rows = d_my_count ? (size_t)mc->cap : (size_t)my_count;
    if (d_my_count) MC_HIP(mc, hipMemcpyAsync(d_cnt + mc->world, d_my_count, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    else {
        hipLaunchKernelGGL(mc_set_count_kernel, dim3(1), dim3(1), 0, st, d_cnt + mc->world, (int32_t)my_count);
        MC_HIP(mc, hipGetLastError());
    }
    if (!mc->comm) {                                             // one rank without a communicator, or a rehearsal group
        uint8_t* slot = arena + (size_t)mc->rank * block;
        if (rows > 0 && d_my_desc != slot)
            MC_HIP(mc, hipMemcpyAsync(slot, d_my_desc, rows * CLC_DESC_BYTES, hipMemcpyDeviceToDevice, st));
        MC_HIP(mc, hipMemcpyAsync(d_cnt + mc->rank, d_cnt + mc->world, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    } else if (mode == CLC_MC_RCCL) {
        // the block is gathered at its fixed capacity (rows >= my_count are padding): d_my_desc must hold maxkp rows
        MC_NCCL(mc, rccl().AllGather(d_my_desc, arena, block, ncclUint8, mc->comm, st));
        MC_NCCL(mc, rccl().AllGather(d_cnt + mc->world, d_cnt, sizeof(int32_t), ncclUint8, mc->comm, st));
    } else {
        const int rc = open_peers(mc, st);
        if (rc != CLC_OK) return rc;
        for (int k = 0; k < mc->world; ++k) {                    // start with the right-hand neighbour: every link busy at once
            const int p = (mc->rank + k) % mc->world;
            if (rows > 0)
                MC_HIP(mc, hipMemcpyAsync(mc->peer_arena[(size_t)p] + (size_t)b * buf + (size_t)mc->rank * block, d_my_desc,
                                          rows * CLC_DESC_BYTES, hipMemcpyDeviceToDevice, st));
        }
        // fence: a rank's contribution to this collective is enqueued behind its copies on the same stream (and behind its
        // sweep of the previous step: see the ordering contract at the top of this file)
        MC_NCCL(mc, rccl().AllGather(d_cnt + mc->world, d_cnt, sizeof(int32_t), ncclUint8, mc->comm, st));
    }
    // the host mirror of the counts travels behind the exchange; whoever needs it synchronises the stream first
    MC_HIP(mc, hipMemcpyAsync(h_cnt, d_cnt, sizeof(int32_t) * (size_t)mc->world, hipMemcpyDeviceToHost, st));
    if (mc->overlap) { MC_HIP(mc, hipEventRecord(mc->ev_exch[b], st)); mc->exch_step[b] = step; }
    mc->cur = b;
    mc->fill = (b + 1) % mc->nbuf;
    mc->steps = step + 1;
    return CLC_OK;
}

int clc_mc_gather_dev(clc_mc* mc, const void* d_my_desc, int my_count, int mode, int* h_counts_out, void* stream)
{
    if (!mc || my_count < 0 || my_count > mc->cap || (my_count > 0 && !d_my_desc) || (mode != CLC_MC_RCCL && mode != CLC_MC_PEER_COPY))
        return mc_fail(mc, CLC_ERR_BAD_ARG, "mc_gather: bad argument");
    MC_HIP(mc, hipSetDevice(mc->device));
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)clc_stream(mc->ctx);
    const int rc = mc_exchange(mc, d_my_desc, my_count, nullptr, mode, st);
    if (rc != CLC_OK) return rc;
    MC_HIP(mc, hipStreamSynchronize(st));                          // clc_mc_match_dev plans its shares from the counts
    mc->counts_on_host = true;
    const int32_t* h_cnt = mc->h_counts + (size_t)mc->cur * (size_t)(mc->world + 1);
    if (h_counts_out) for (int c = 0; c < mc->world; ++c) h_counts_out[c] = h_cnt[c];
    return CLC_OK;
}

int clc_mc_gather_enqueue_dev(clc_mc* mc, const void* d_my_desc, int my_count, const int32_t* d_my_count, int mode, void* stream)
{
    if (!mc || !d_my_desc || (!d_my_count && (my_count < 0 || my_count > mc->cap)) || (mode != CLC_MC_RCCL && mode != CLC_MC_PEER_COPY))
        return mc_fail(mc, CLC_ERR_BAD_ARG, "mc_gather_enqueue: bad argument");
    MC_HIP(mc, hipSetDevice(mc->device));
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)clc_stream(mc->ctx);
    mc->counts_on_host = false;
    return mc_exchange(mc, d_my_desc, my_count, d_my_count, mode, st);
}

int clc_mc_open_peers(clc_mc* mc, void* stream)
{
    if (!mc) return CLC_ERR_BAD_ARG;
    if (!mc->comm && mc->world > 1) return mc_fail(mc, CLC_ERR_STATE, "mc_open_peers: rehearsal handle (no communicator)");
    MC_HIP(mc, hipSetDevice(mc->device));
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)clc_stream(mc->ctx);
    return open_peers(mc, st);
}

int clc_mc_counts(clc_mc* mc, int* h_counts_out, void* stream)
{
    if (!mc || !h_counts_out) return mc_fail(mc, CLC_ERR_BAD_ARG, "mc_counts: bad argument");
    MC_HIP(mc, hipSetDevice(mc->device));
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)clc_stream(mc->ctx);
    MC_HIP(mc, hipStreamSynchronize(st));
    const int32_t* h_cnt = mc->h_counts + (size_t)mc->cur * (size_t)(mc->world + 1);
    for (int c = 0; c < mc->world; ++c) h_counts_out[c] = h_cnt[c];
    return CLC_OK;
}

int clc_mc_virtual_put(clc_mc* mc, int other_rank, const void* d_desc, int count, void* stream)
{
    if (!mc || !mc->is_virtual || other_rank < 0 || other_rank >= mc->world || count < 0 || count > mc->cap || (count > 0 && !d_desc))
        return mc_fail(mc, CLC_ERR_BAD_ARG, "mc_virtual_put: bad argument (only for handles created without an id)");
    MC_HIP(mc, hipSetDevice(mc->device));
    hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)clc_stream(mc->ctx);
    // plays another rank's part of the NEXT exchange: its block and count go where that rank's copy / collective would put them
    const size_t block = (size_t)mc->cap * CLC_DESC_BYTES;
    if (mc->overlap && mc->sweep_step[mc->fill] >= 0) MC_HIP(mc, hipStreamWaitEvent(st, mc->ev_sweep[mc->fill], 0));   // (a real peer's copy comes behind the collective chain)
    if (count > 0)
        MC_HIP(mc, hipMemcpyAsync(mc->d_arena + ((size_t)mc->fill * (size_t)mc->world + (size_t)other_rank) * block, d_desc,
                                  (size_t)count * CLC_DESC_BYTES, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(mc_set_count_kernel, dim3(1), dim3(1), 0, st, mc->d_counts + (size_t)mc->fill * (size_t)(mc->world + 1) + other_rank,
                       (int32_t)count);
    MC_HIP(mc, hipGetLastError());
    MC_HIP(mc, hipStreamSynchronize(st));                        // a rehearsal entry: on return the "peer's" block has arrived
    return CLC_OK;
}

// shares -> jobs over buffer `cur`; device_counts: the jobs read the gathered counts themselves (capacity-planned shares)
static int mc_sweep(clc_mc* mc, const std::vector<int>& counts, bool device_counts, int threshold, int32_t* d_match, int match_capacity,
                    clc_mc_share* h_shares, int share_capacity, int* n_shares, void* stream)
{
    const int grain = clc_k2nn_queries_per_block(mc->ctx);
    std::vector<clc_mc_share> shares((size_t)mc->world * (size_t)mc->world + 2);
    int n = 0;
    int rc = clc_mc_plan(counts.data(), mc->world, mc->world, mc->rank, grain, shares.data(), (int)shares.size(), &n);
    if (rc != CLC_OK) return mc_fail(mc, rc, "mc_match: planning failed");
    if (n > share_capacity) return mc_fail(mc, CLC_ERR_CAPACITY, "mc_match: share array too small");
    std::vector<clc_match_job> jobs((size_t)n);
    std::vector<const int32_t*> cq, ct;
    const int32_t* d_cnt = mc->d_counts + (size_t)mc->cur * (size_t)(mc->world + 1);
    uint64_t rows = 0;
    for (int k = 0; k < n; ++k) {
        const clc_mc_share& s = shares[(size_t)k];
        jobs[(size_t)k].q_offset = (uint32_t)s.first * (uint32_t)mc->cap + s.q_begin;
        jobs[(size_t)k].nq = s.nq;
        jobs[(size_t)k].t_offset = (uint32_t)s.second * (uint32_t)mc->cap;
        jobs[(size_t)k].nt = (uint32_t)counts[(size_t)s.second];
        jobs[(size_t)k].out_offset = s.out_offset;
        jobs[(size_t)k].threshold = (uint32_t)threshold;
        rows += s.nq;
        if (h_shares) h_shares[k] = s;
        if (device_counts) { cq.push_back(d_cnt + s.first); ct.push_back(d_cnt + s.second); }
    }
    if (rows > (uint64_t)match_capacity) return mc_fail(mc, CLC_ERR_CAPACITY, "mc_match: result buffer too small");
    *n_shares = n;
    if (n == 0) return CLC_OK;
    const uint8_t* arena = mc->d_arena + (size_t)mc->cur * (size_t)mc->world * (size_t)mc->cap * CLC_DESC_BYTES;
    hipStream_t sweep_st = stream ? (hipStream_t)stream : (hipStream_t)clc_stream(mc->ctx);
    if (mc->overlap && mc->exch_step[mc->cur] >= 0) MC_HIP(mc, hipStreamWaitEvent(sweep_st, mc->ev_exch[mc->cur], 0));
    if (device_counts) {
        std::vector<uint32_t> row0((size_t)n);
        for (int k = 0; k < n; ++k) row0[(size_t)k] = shares[(size_t)k].q_begin;
        rc = clc_match_jobs_counted_dev(mc->ctx, arena, jobs.data(), n, cq.data(), ct.data(), row0.data(), d_match, stream);
    } else rc = clc_match_jobs_dev(mc->ctx, arena, jobs.data(), n, d_match, stream);
    if (rc != CLC_OK) return mc_fail(mc, rc, clc_last_error_string(mc->ctx));
    if (mc->overlap) { MC_HIP(mc, hipEventRecord(mc->ev_sweep[mc->cur], sweep_st)); mc->sweep_step[mc->cur] = mc->exch_step[mc->cur]; }
    return CLC_OK;
}

int clc_mc_match_dev(clc_mc* mc, int threshold, int32_t* d_match, int match_capacity, clc_mc_share* h_shares, int share_capacity,
                     int* n_shares, void* stream)
{
    if (!mc || !n_shares || match_capacity < 0 || (match_capacity > 0 && !d_match) || share_capacity < 0 || (share_capacity > 0 && !h_shares))
        return mc_fail(mc, CLC_ERR_BAD_ARG, "mc_match: bad argument");
    if (!mc->counts_on_host) return mc_fail(mc, CLC_ERR_STATE, "mc_match: the last exchange was enqueue-only (use clc_mc_match_enqueue_dev, or clc_mc_gather_dev)");
    std::vector<int> counts((size_t)mc->world);
    const int32_t* h_cnt = mc->h_counts + (size_t)mc->cur * (size_t)(mc->world + 1);
    for (int c = 0; c < mc->world; ++c) counts[(size_t)c] = h_cnt[c];
    return mc_sweep(mc, counts, false, threshold, d_match, match_capacity, h_shares, share_capacity, n_shares, stream);
}

int clc_mc_match_enqueue_dev(clc_mc* mc, int threshold, int32_t* d_match, int match_capacity, clc_mc_share* h_shares, int share_capacity,
                             int* n_shares, void* stream)
{
    if (!mc || !n_shares || match_capacity < 0 || (match_capacity > 0 && !d_match) || share_capacity < 0 || (share_capacity > 0 && !h_shares))
        return mc_fail(mc, CLC_ERR_BAD_ARG, "mc_match_enqueue: bad argument");
    // the shares are cut on the block CAPACITY (the same for every step of a handle); the sweep reads the gathered counts from
    // device memory, answers -1 for planned query rows past a camera's count and sweeps only the train rows that exist
    const std::vector<int> counts((size_t)mc->world, mc->cap);
    return mc_sweep(mc, counts, true, threshold, d_match, match_capacity, h_shares, share_capacity, n_shares, stream);
}

} // extern "C"
