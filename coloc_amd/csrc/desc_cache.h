// desc_cache.h -- detector -> matcher hand-over of descriptor blocks without a second upload (internal; not installed).
//
// The reference passes descriptors from GPUDetector to GPUMatcher through host memory (FeatureMap regions, GPUDetector.hpp:181 ->
// GPUMatcher.hpp:188-196) and uploads them again in every match call.  Here the front end writes a frame's descriptors into a device
// block OWNED BY THIS TABLE (desc_reserve), and when the host stores the rows at a host address it PUBLISHES that fact (desc_publish):
// address + count + a generation stamp (+ a 64-bit fold of all rows).  A host-pointer match entry that is later given that address and
// count finds the rows on the device.  What makes a hit safe is decided by the looking-up context's mode (clc_desc_cache_mode):
//   VERIFY (default everywhere, the policy classes included): the whole host block is folded again and compared with the fold taken
//           at publish time -- WHILE the GPU already sweeps the device rows (the caller enqueues first and verifies behind it); a block
//           edited anywhere is uploaded and the sweep repeated, so edited host rows are never matched stale;
//   TRUST:  address, count, generation and 18 sampled rows; the integrator's statement that published blocks are not edited in place;
//   OFF:    every block is uploaded, as the reference does.
// An entry dies when its owner publishes the same host address again, when the owner context is destroyed (freeGPUMemory), when a
// lookup sees changed rows, or when it is the least recently used of 32.
#ifndef CLC_DESC_CACHE_H
#define CLC_DESC_CACHE_H

#include <stddef.h>
#include <stdint.h>

#include "../../include/coloc_hip.h"

namespace clc {

struct DescEntry;

// 64-bit position-keyed multiply-fold of n descriptor rows (change detection, not cryptography); the copying form writes dst while it
// folds src (one pass: what the front end's single copy into a regions block costs anyway).
uint64_t desc_block_fold(const void* h, size_t n);
uint64_t desc_copy_fold(void* dst, const void* src, size_t n);

// A device block of >= rows rows on `device` for `owner` to write a frame's descriptors into; unpublished until desc_publish.
// nullptr when every entry is in use (the caller then writes into memory of its own and nothing is published).
DescEntry* desc_reserve(const clc_ctx* owner, int device, size_t rows, uint8_t** d_rows);
uint8_t* desc_rows(DescEntry* e);
void desc_abandon(DescEntry* e);      // a reservation that will not be published after all
// The n rows of entry e are the rows now stored at host address h.  fold: desc_block_fold of them when has_fold.
void desc_publish(DescEntry* e, const void* h, int n, uint64_t fold, bool has_fold, clc_desc_handle* out);
// Device rows of host block (h, n) if a live entry stands for it under `mode`; the entry stays pinned until desc_release.
// *needs_verify: the caller must call desc_verify (after enqueueing its device work) and repeat with uploaded rows if it says false.
const uint8_t* desc_acquire(int mode, int device, const void* h, int n, DescEntry** held, bool* needs_verify);
bool desc_verify(DescEntry* held, const void* h, int n);
void desc_release(DescEntry* held);
// freeGPUMemory of the owner: its entries die (their device blocks are freed unless a running call still reads them)
void desc_drop_owner(const clc_ctx* owner);

} // namespace clc
#endif
