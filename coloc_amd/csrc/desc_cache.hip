// desc_cache.hip -- the table behind desc_cache.h and the clc_desc_cache_* entry points (include/coloc_hip.h).
// Replaces nothing in the reference, which uploads both descriptor sets in every match call (GPUMatcher.hpp:188-196).
#include "clc_ctx.h"
#include "desc_cache.h"

#include <cstring>
#include <mutex>
#include <vector>

namespace clc {

enum { kCacheSamples = 16 };
struct DescEntry {
    int device = -1;
    const clc_ctx* owner = nullptr;   // the context that reserved the block (identity only, never dereferenced)
    const void* h = nullptr;          // host address the rows were published at; nullptr: unpublished / dead
    int n = 0;
    uint64_t generation = 0;          // stamped at publish time from a process-wide counter
    uint8_t first[CLC_DESC_BYTES], last[CLC_DESC_BYTES];
    uint8_t sample[kCacheSamples][CLC_DESC_BYTES];      // rows (i + 1) * n / (kCacheSamples + 1)
    uint64_t fold = 0;
    bool has_fold = false;
    uint8_t* d = nullptr;
    size_t cap = 0;                   // rows allocated
    uint64_t stamp = 0;               // last use
    int busy = 0;                     // running calls that read d (or a reservation being filled)
    bool reserved = false;            // handed to a front end that has not published it yet
};
struct DescCache {
    std::mutex mu;
    std::vector<DescEntry> e;
    uint64_t clock = 0, generation = 0;
    uint64_t hits = 0, misses = 0, rejected = 0;        // lookups answered from the table / uploaded / found changed by a verification
};
static constexpr size_t kDescCacheEntries = 32;
static DescCache& table() { static DescCache c; return c; }

static inline uint64_t fold_mul(const uint64_t a, const uint64_t b)
{
    const unsigned __int128 p = (unsigned __int128)a * b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}
// every 16 bytes are keyed with their position (the key steps per row, the four pairs of a row take different constants), multiplied
// 64 x 64 -> 128 and folded; the row terms are summed.  One multiply per 16 bytes: a block takes what reading it takes.
template <bool COPY>
static uint64_t fold_rows(void* dst, const void* src, const size_t n)
{
    const uint8_t* p = (const uint8_t*)src;
    uint8_t* o = (uint8_t*)dst;
    uint64_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, k = 0x9E3779B97F4A7C15ull;
    for (size_t r = 0; r < n; ++r, p += CLC_DESC_BYTES, k += 0xD1B54A32D192ED03ull) {
        uint64_t w[8];
        __builtin_prefetch(p + 8 * CLC_DESC_BYTES);     // (a block the GPU has just written is not in any cache: ask eight rows ahead)
        memcpy(w, p, sizeof w);                          // (host blocks carry no alignment promise)
        if (COPY) { memcpy(o, w, sizeof w); o += CLC_DESC_BYTES; }
        h0 += fold_mul(w[0] ^ k, w[1] ^ 0x8BB84B93962EACC9ull);
        h1 += fold_mul(w[2] ^ (k + 0x2D358DCCAA6C78A5ull), w[3] ^ 0x4B33A62ED433D4A3ull);
        h2 += fold_mul(w[4] ^ (k + 0x4D5A2DA51DE1AA47ull), w[5] ^ 0xA0761D6478BD642Full);
        h3 += fold_mul(w[6] ^ (k + 0xE7037ED1A0B428DBull), w[7] ^ 0x589965CC75374CC3ull);
    }
    return (h0 ^ (h1 << 1 | h1 >> 63)) + (h2 ^ (h3 << 7 | h3 >> 57)) + (uint64_t)n;
}
uint64_t desc_block_fold(const void* h, const size_t n) { return fold_rows<false>(nullptr, h, n); }
uint64_t desc_copy_fold(void* dst, const void* src, const size_t n) { return fold_rows<true>(dst, src, n); }

static inline const uint8_t* sample_row(const void* h, const int n, const int i)
{
    return (const uint8_t*)h + (size_t)((uint64_t)(i + 1) * (uint64_t)n / (kCacheSamples + 1)) * CLC_DESC_BYTES;
}

DescEntry* desc_reserve(const clc_ctx* owner, const int device, const size_t rows, uint8_t** d_rows)
{
    *d_rows = nullptr;
    DescCache& c = table();
    DescEntry* slot = nullptr;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        // a block this owner reserved and never published (or whose publication has died) serves again
        for (DescEntry& en : c.e)
            if (en.owner == owner && en.device == device && !en.h && en.busy == 0 && en.cap >= rows) { slot = &en; break; }
        if (!slot && c.e.size() < kDescCacheEntries) {
            c.e.reserve(kDescCacheEntries);                                          // entries never move (held pointers)
            c.e.emplace_back();
            slot = &c.e.back();
        }
        if (!slot)
            for (DescEntry& en : c.e) {                                               // a dead entry first, else the least recently used
                if (en.busy != 0 || en.reserved) continue;
                if (!slot || (!en.h && slot->h) || (!en.h == !slot->h && en.stamp < slot->stamp)) slot = &en;
            }
        if (!slot) return nullptr;                                                    // everything in use
        slot->busy = 1;                                                               // while it is being (re)allocated
        slot->h = nullptr; slot->n = 0; slot->has_fold = false;
        slot->reserved = true;
        slot->owner = owner;
        slot->stamp = ++c.clock;
    }
    hipError_t e = hipSuccess;
    if (slot->cap < rows || slot->device != device) {
        if (slot->d) { (void)hipSetDevice(slot->device); (void)hipFree(slot->d); (void)hipSetDevice(device); }
        slot->d = nullptr; slot->cap = 0;
        e = hipMalloc((void**)&slot->d, rows * CLC_DESC_BYTES);
        if (e == hipSuccess) slot->cap = rows;
    }
    std::lock_guard<std::mutex> lk(c.mu);
    slot->busy = 0;
    slot->device = device;
    if (e != hipSuccess) { slot->reserved = false; slot->owner = nullptr; return nullptr; }
    *d_rows = slot->d;
    return slot;
}

uint8_t* desc_rows(DescEntry* e) { return e ? e->d : nullptr; }

void desc_abandon(DescEntry* e)
{
    if (!e) return;
    std::lock_guard<std::mutex> lk(table().mu);
    e->reserved = false; e->h = nullptr; e->owner = nullptr;
}

void desc_publish(DescEntry* e, const void* h, const int n, const uint64_t fold, const bool has_fold, clc_desc_handle* out)
{
    DescCache& c = table();
    std::lock_guard<std::mutex> lk(c.mu);
    // whatever stood for this host address before no longer does (a new frame's regions block often lands where the last one lay)
    for (DescEntry& en : c.e)
        if (&en != e && en.device == e->device && en.h == h) en.h = nullptr;
    e->h = h; e->n = n; e->fold = fold; e->has_fold = has_fold;
    e->generation = ++c.generation;
    e->stamp = ++c.clock;
    e->reserved = false;
    memcpy(e->first, h, CLC_DESC_BYTES);
    memcpy(e->last, (const uint8_t*)h + (size_t)(n - 1) * CLC_DESC_BYTES, CLC_DESC_BYTES);
    for (int i = 0; i < kCacheSamples; ++i) memcpy(e->sample[i], sample_row(h, n, i), CLC_DESC_BYTES);
    if (out) { out->host = h; out->count = (uint32_t)n; out->slot = (uint32_t)(e - c.e.data()); out->generation = e->generation; }
}

const uint8_t* desc_acquire(const int mode, const int device, const void* h, const int n, DescEntry** held, bool* needs_verify)
{
    *held = nullptr;
    *needs_verify = false;
    if (mode == CLC_DESC_CACHE_OFF || !h || n <= 0) return nullptr;
    DescCache& c = table();
    std::lock_guard<std::mutex> lk(c.mu);
    for (DescEntry& en : c.e) {
        if (en.device != device || en.h != h || en.n != n || !en.d) continue;
        if (mode == CLC_DESC_CACHE_VERIFY && !en.has_fold) continue;        // published without a fold: not for a verifying context
        // 18 rows first: a block that was rewritten wholesale (an allocation reused for other rows) is turned away here, before any
        // device work is enqueued on its old rows
        const uint8_t* hb = (const uint8_t*)h;
        bool same = memcmp(hb, en.first, CLC_DESC_BYTES) == 0 && memcmp(hb + (size_t)(n - 1) * CLC_DESC_BYTES, en.last, CLC_DESC_BYTES) == 0;
        for (int i = 0; same && i < kCacheSamples; ++i) same = memcmp(sample_row(h, n, i), en.sample[i], CLC_DESC_BYTES) == 0;
        if (!same) { en.h = nullptr; ++c.rejected; continue; }
        en.stamp = ++c.clock;
        ++en.busy;
        *held = &en;
        *needs_verify = mode == CLC_DESC_CACHE_VERIFY;
        if (!*needs_verify) ++c.hits;                                        // (a verifying lookup counts when the fold has agreed)
        return en.d;
    }
    ++c.misses;
    return nullptr;
}

bool desc_verify(DescEntry* held, const void* h, const int n)
{
    if (!held) return false;
    const uint64_t fold = desc_block_fold(h, (size_t)n);                     // the pass over the block, outside the lock
    DescCache& c = table();
    std::lock_guard<std::mutex> lk(c.mu);
    const bool ok = held->has_fold && held->h == h && held->n == n && fold == held->fold;
    if (ok) ++c.hits;
    else { if (held->h == h) held->h = nullptr; ++c.rejected; ++c.misses; }
    return ok;
}

void desc_release(DescEntry* held)
{
    if (!held) return;
    std::lock_guard<std::mutex> lk(table().mu);
    --held->busy;
}

void desc_drop_owner(const clc_ctx* owner)
{
    DescCache& c = table();
    std::lock_guard<std::mutex> lk(c.mu);
    for (DescEntry& en : c.e) {
        if (en.owner != owner) continue;
        en.owner = nullptr; en.h = nullptr; en.n = 0; en.reserved = false;
        if (en.busy == 0 && en.d) { (void)hipSetDevice(en.device); (void)hipFree(en.d); en.d = nullptr; en.cap = 0; }
    }
}

} // namespace clc

using namespace clc;

extern "C" {

int clc_desc_cache_mode(clc_ctx* ctx, int mode)
{
    if (!ctx || (mode != CLC_DESC_CACHE_OFF && mode != CLC_DESC_CACHE_VERIFY && mode != CLC_DESC_CACHE_TRUST))
        return fail(ctx, CLC_ERR_BAD_ARG, "desc_cache_mode: unknown mode");
    ctx->cache_mode = mode;
    return CLC_OK;
}

int clc_desc_cache_publish(clc_ctx* ctx, const void* d_src, const void* h_desc, int n, clc_desc_handle* handle)
{
    if (handle) memset(handle, 0, sizeof *handle);
    if (!ctx || !h_desc || n < 0) return fail(ctx, CLC_ERR_BAD_ARG, "desc_cache_publish: bad argument");
    if (ctx->cache_mode == CLC_DESC_CACHE_OFF || n == 0) return CLC_OK;
    CLC_HIP(ctx, hipSetDevice(ctx->device));
    DescEntry* e = nullptr;
    if (!d_src) {
        if (!ctx->has_det) return fail(ctx, CLC_ERR_STATE, "desc_cache_publish: context created without detector options and no device source given");
        // the rows of this context's last clc_detect_and_describe* call already lie in a block of the table ...
        if (ctx->desc_pending && n == ctx->staged_n) e = ctx->desc_pending;
        else if (ctx->h_stage && ctx->staged_n >= 0 && n <= ctx->staged_n) {
            // ... unless that block has been published already (the same frame stored at a second address): the rows are still in
            // the pinned staging block
            uint8_t* d = nullptr;
            e = desc_reserve(ctx, ctx->device, (size_t)n, &d);
            if (!e) return CLC_OK;
            hipError_t err = hipMemcpyAsync(d, ctx->h_stage + ctx->stage_desc, (size_t)n * CLC_DESC_BYTES, hipMemcpyHostToDevice, ctx->stream);
            if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
            if (err != hipSuccess) { desc_abandon(e); return fail(ctx, CLC_ERR_HIP, "desc_cache_publish", err); }
        } else d_src = ctx->d_desc;           // rows of the device-resident flow (clc_detect_dev + clc_describe_detected_dev)
    }
    if (!e) {
        uint8_t* d = nullptr;
        e = desc_reserve(ctx, ctx->device, (size_t)n, &d);
        if (!e) return CLC_OK;                                                       // every entry in use: nothing is published
        hipError_t err = hipMemcpyAsync(d, d_src, (size_t)n * CLC_DESC_BYTES, hipMemcpyDeviceToDevice, ctx->stream);
        if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
        if (err != hipSuccess) { desc_abandon(e); return fail(ctx, CLC_ERR_HIP, "desc_cache_publish", err); }
    }
    // a verifying context folds the block it publishes (the host has just written it); a trusting one does not, and its entries can
    // then only be hit by trusting lookups
    const bool with_fold = ctx->cache_mode == CLC_DESC_CACHE_VERIFY;
    desc_publish(e, h_desc, n, with_fold ? desc_block_fold(h_desc, (size_t)n) : 0u, with_fold, handle);
    if (e == ctx->desc_pending) ctx->desc_pending = nullptr;                         // the next frame takes another block
    return CLC_OK;
}

int clc_desc_handle_live(const clc_desc_handle* handle)
{
    if (!handle || !handle->host) return 0;
    DescCache& c = table();
    std::lock_guard<std::mutex> lk(c.mu);
    if (handle->slot >= c.e.size()) return 0;
    const DescEntry& en = c.e[handle->slot];
    return en.h == handle->host && en.generation == handle->generation && (uint32_t)en.n == handle->count && en.d ? 1 : 0;
}

int clc_desc_cache_stats(unsigned long long* hits, unsigned long long* misses)
{
    DescCache& c = table();
    std::lock_guard<std::mutex> lk(c.mu);
    if (hits) *hits = c.hits;
    if (misses) *misses = c.misses;
    return CLC_OK;
}

int clc_desc_cache_clear(void)
{
    DescCache& c = table();
    std::lock_guard<std::mutex> lk(c.mu);
    for (DescEntry& en : c.e) {
        if (en.busy || en.reserved) { en.h = nullptr; continue; }
        if (en.d) { (void)hipSetDevice(en.device); (void)hipFree(en.d); }
        en.d = nullptr; en.cap = 0; en.h = nullptr; en.n = 0; en.owner = nullptr;
    }
    return CLC_OK;
}

} // extern "C"
