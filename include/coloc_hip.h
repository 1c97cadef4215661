/*
 * coloc_hip.h -- C ABI of libcoloc_hip.so: the MI355X (gfx950) implementation of CoLoC's
 * describe -> 512-bit Hamming 2-NN match -> batched PnP scoring hot path.
 *
 * This is the drop-in boundary.  It replaces the reference's static library `koral`
 * (CMakeLists.txt:37 = src/CUDALERP.cu + src/CLATCH.cu + src/CUDAK2NN.cu) and the CUDA-runtime
 * plumbing inside include/coloc/GPUDetector.hpp / GPUMatcher.hpp.  Plain C types only: no C++
 * types, no exceptions, no exit() -- every entry point returns an int status (the reference
 * aborts via exit(code) on four calls and ignores all other CUDA errors,
 * GPUMatcher.hpp:33-41,188-195).
 *
 * Conventions
 *   - descriptors: rows of 64 bytes = 16 little-endian uint32 = 8 little-endian uint64
 *     (CLATCH.cu:185-188, CUDAK2NN.cu:47-52).
 *   - `h_` pointers are host memory (caller-owned), `d_` pointers are device memory on the
 *     context's GPU (caller-owned unless stated); `stream` is a hipStream_t passed as void*
 *     (NULL = the context's own stream).  Host-pointer entry points are synchronous on return,
 *     like the reference (CUDAK2NN.cu:80, GPUDetector.hpp:290); `_dev` entry points only enqueue.
 *   - device buffers owned by the context are sized from the options' maxkp with an explicit
 *     capacity check (CLC_ERR_CAPACITY); the reference has none (GPUDetector.hpp:135,281).
 *   - inputs need no padding (the reference needs +8 readable train vectors,
 *     GPUMatcher.hpp:183-186).
 *   - a context is thread-compatible: one host thread at a time; several contexts per
 *     process / device are fine; no hidden process-global state.  A context owns ONE set of device
 *     workspaces (pyramid arena, matcher top-2 rows and arrival counters, pose scratch), so the `_dev`
 *     calls made on one context must be ordered with respect to each other -- same stream, or
 *     streams joined by events; use one context per concurrently running stream.
 */
#ifndef COLOC_HIP_H
#define COLOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: rounds 1-3.  2: round 4's additions (clc_detect_batch_dev, clc_desc_cache_*, clc_k2nn_plan_query, clc_mc_open_peers,
 * clc_pnp_localize_ac_batch / clc_pose_job, one-rank communicators in clc_mc_create).  3: round 5's additions (clc_desc_cache_mode,
 * clc_describe_match_pair_dev, clc_essential_acransac_batch, clc_inter_pose_batch, clc_k2nn_device_info) and round 6's changes
 * (clc_describe_match_pair_dev lost its `chunks` argument, CLC_K2NN_MATRIX_PLAIN is gone, the descriptor hand-over is by ownership:
 * clc_desc_cache_publish returns a handle).  4: the 'F' / 'H' models of the two-view filter (clc_two_view_acransac, _batch, clc_two_view_minimal).
 * Bindings check clc_abi_version() BEFORE resolving symbols an older library does not export. */
#define CLC_ABI_VERSION 4
#define CLC_DESC_BYTES 64
#define CLC_MAX_LEVELS 8
#define CLC_MAX_BATCH 8    /* cameras per clc_describe_batch_dev / clc_detect_batch_dev call */
#define CLC_DETECT_MAX_WIDTH 4096   /* widest image the GPU detector takes (its row-walk replay keeps a row's pre-test bits in LDS) */

/* status codes */
enum {
    CLC_OK = 0,
    CLC_ERR_BAD_ARG = 1,    /* null pointer, negative count, misaligned device pointer ...      */
    CLC_ERR_CAPACITY = 2,   /* more keypoints / descriptors / hypotheses than the ctx was sized for */
    CLC_ERR_HIP = 3,        /* a HIP runtime call failed; see clc_last_error_string            */
    CLC_ERR_NO_DEVICE = 4,  /* no usable gfx950 device                                          */
    CLC_ERR_STATE = 5       /* call order violated (e.g. describe before pyramid_build)        */
};

/* mirrors coloc::DetectorOptions, include/coloc/colocData.hpp:29-36 */
typedef struct clc_detector_opts {
    float    scale_factor;  /* 1.2f in coloc_node.cpp:77 */
    uint8_t  scale_levels;  /* 8; at most CLC_MAX_LEVELS (CLATCH indexes d_all_tex[8], CLATCH.cu:160) */
    uint32_t width;
    uint32_t height;
    uint32_t maxkp;
    uint8_t  thresh;        /* FAST threshold (host feeder), coloc_node.cpp:81 */
} clc_detector_opts;

/* mirrors coloc::MatcherOptions, include/coloc/colocData.hpp:38-42 */
typedef struct clc_matcher_opts {
    float    distRatio;     /* unused by the 2-NN difference rule; kept for the CPU comparator */
    int      thresh;        /* map-tracking threshold (Mopts.thresh = 60, coloc_node.cpp:85)    */
    uint32_t maxkp;
} clc_matcher_opts;

/* wire format shared with the host feeder: include/coloc/Keypoint.h:155-163, sizeof == 20 */
typedef struct clc_keypoint {
    int32_t x;      /* level-local pixel column */
    int32_t y;      /* level-local pixel row    */
    uint8_t score;
    float   angle;  /* radians                  */
    uint8_t scale;  /* pyramid level 0..7       */
} clc_keypoint;

typedef struct clc_ctx clc_ctx;

/* one unit of matcher work for clc_match_jobs_dev: queries [q_begin, q_begin+nq) of a query set
 * against a whole train set.  Offsets are in descriptors (rows of 64 B) from the base pointers. */
typedef struct clc_match_job {
    uint32_t q_offset;   /* first query row, relative to d_desc_base        */
    uint32_t nq;
    uint32_t t_offset;   /* first train row, relative to d_desc_base        */
    uint32_t nt;
    uint32_t out_offset; /* first int32 of the result, relative to d_match  */
    uint32_t threshold;  /* truncated to 8 bits, see clc_match_2nn          */
} clc_match_job;

/* descriptor hand-over (see clc_desc_cache_mode below) */
/* what a publication hands out: the host address and row count it stands for, and the generation it was stamped with (a process-wide
 * counter: publishing the same address again, destroying the publishing context -- freeGPUMemory -- or a lookup that sees changed
 * rows ends a generation).  clc_desc_handle_live: 1 while that very publication still stands. */
typedef struct clc_desc_handle {
    const void* host;
    uint64_t    generation;
    uint32_t    count;
    uint32_t    slot;
} clc_desc_handle;

/* ---- lifecycle --------------------------------------------------------------------------- */

int clc_abi_version(void);
const char* clc_status_string(int status);

/* Replaces the constructors GPUDetector(DetectorOptions) (GPUDetector.hpp:70-138) and
 * GPUMatcher(MatcherOptions) (GPUMatcher.hpp:70-95): allocates pyramid levels, keypoint /
 * descriptor / match buffers and the learned triplet table on device `device_id`.
 * Either opts pointer may be NULL if that half is not used. */
int clc_ctx_create(int device_id, const clc_detector_opts* dopts, const clc_matcher_opts* mopts,
                   clc_ctx** out_ctx);
/* Replaces freeGPUMemory() (GPUDetector.hpp:144-155, GPUMatcher.hpp:102-108). */
int clc_ctx_destroy(clc_ctx* ctx);
const char* clc_last_error_string(const clc_ctx* ctx);
/* Blocks until everything enqueued on the context's stream has finished. */
int clc_sync(clc_ctx* ctx);
/* The context's hipStream_t (as void*), for callers that enqueue their own work in order. */
void* clc_stream(clc_ctx* ctx);
/* The device the context lives on. */
int clc_ctx_device(const clc_ctx* ctx);

/* ---- kernel timing (replaces the std::chrono prints around CUDAK2NN, GPUMatcher.hpp:204-206) ---
 * When enabled, every kernel launch made through this context is bracketed by a pair of HIP
 * events on the stream it is launched on.  clc_profile_read drains the finished pairs (it
 * synchronises those streams) and returns the accumulated device time and launch count of one
 * kernel since the last clc_profile_reset. */
enum {
    CLC_KERNEL_PYRAMID = 0,
    CLC_KERNEL_CLATCH = 1,
    CLC_KERNEL_K2NN_SWEEP = 2,
    CLC_KERNEL_K2NN_MERGE = 3,
    CLC_KERNEL_PNP_RESIDUALS = 4,
    CLC_KERNEL_PNP_SCORE = 5,
    CLC_KERNEL_DETECT = 6,      /* the FAST-9 / NMS / emit launch group, bracketed as one */
    CLC_KERNEL_COUNT = 7
};
/* on = 0: off; 1: bracket every kernel; otherwise a mask, bit (k + 1) selecting kernel k. */
int clc_profile_enable(clc_ctx* ctx, int on);
int clc_profile_reset(clc_ctx* ctx);
int clc_profile_read(clc_ctx* ctx, int kernel, double* total_ms, int* launches);
const char* clc_kernel_name(int kernel);

/* ---- pyramid: replaces CUDALERP() (CUDALERP.h:166) + the level loop GPUDetector.hpp:232-255 -- */

/* Upload a WxH u8 image (tight rows) as level 0 and resample levels 1..L-1 from it. */
int clc_pyramid_build(clc_ctx* ctx, const uint8_t* h_img, uint32_t width, uint32_t height);
/* Same from a device image with `pitch` bytes per row. */
int clc_pyramid_build_dev(clc_ctx* ctx, const void* d_img, uint32_t width, uint32_t height,
                          size_t pitch, void* stream);
/* Geometry and device address of one level (GPUDetector.hpp:109-114 dims). */
int clc_pyramid_level(const clc_ctx* ctx, int level, uint32_t* w, uint32_t* h, size_t* pitch,
                      const void** d_ptr);
/* Download one level into tight host rows (the reference's per-level D2H, GPUDetector.hpp:265). */
int clc_pyramid_download(clc_ctx* ctx, int level, uint8_t* h_out);

/* ---- detect: replaces the host loop KFAST<true,true> + featureAngle per level ---------------
 * (GPUDetector.hpp:262-277; KFAST.h:502-540; FeatureAngle.h:197-246) -- on the GPU, all levels.
 * Keypoints come out in the reference's order: level-major, then (y, x) ascending, with level-local
 * integer coordinates, corner score, orientation and level.  FAST threshold = opts.thresh.
 * At most DetectorOptions.maxkp keypoints are kept (the first ones in that order; the reference
 * overflows its buffers instead, GPUDetector.hpp:135,281); *n_found reports how many there were. */
int clc_detect(clc_ctx* ctx, clc_keypoint* h_kps, int capacity, int* n_written, int* n_found);
/* Device-resident: keypoints stay in the context (clc_detect_buffers), nothing is copied back. */
int clc_detect_dev(clc_ctx* ctx, void* stream);
/* Device addresses of the context's keypoint array, of its uint32 count pair {written, found} and
 * of its descriptor array (valid for the context's lifetime). */
int clc_detect_buffers(clc_ctx* ctx, const clc_keypoint** d_kps, const uint32_t** d_count, void** d_desc);
/* CLATCH over the context's own keypoints with the count taken from device memory; writes
 * descriptors to d_desc (NULL = the context's descriptor array). */
int clc_describe_detected_dev(clc_ctx* ctx, void* d_desc, void* stream);
/* Whole front end on host buffers, like GPUDetector::detectAndDescribe (GPUDetector.hpp:216-291):
 * upload image -> pyramid -> detect -> describe -> download keypoints + descriptors. */
int clc_detect_and_describe(clc_ctx* ctx, const uint8_t* h_img, uint32_t width, uint32_t height,
                            clc_keypoint* h_kps, uint8_t* h_desc, int capacity, int* n_written, int* n_found);
/* The same without the copies out: *h_kps / *h_desc point into the context's pinned staging block (valid until the next call of this
 * family on the context), *n_written keypoints / rows of 64 B.  ONE enqueue sequence and ONE stream synchronisation per frame: image
 * into pinned memory -> DMA upload -> pyramid -> detector -> CLATCH -> one small launch that mirrors the count, the keypoints and the
 * descriptors that were found into the pinned block (the reference: 7 level downloads each with a synchronisation, two uploads, a
 * download, a device synchronisation, GPUDetector.hpp:262-290). */
int clc_detect_and_describe_view(clc_ctx* ctx, const uint8_t* h_img, uint32_t width, uint32_t height, const clc_keypoint** h_kps,
                                 const uint8_t** h_desc, int* n_written, int* n_found);
/* The frame's ONE copy of its descriptors into the caller's block (regions[idx]->Descriptors(), GPUDetector.hpp:181): copies the first n
 * rows staged by the last clc_detect_and_describe* call to h_dst and -- when n is all of them and the context's cache mode is not OFF --
 * publishes h_dst (see clc_desc_cache_mode): the rows are on the device already, the fold is taken during the copy. */
int clc_detect_store_descriptors(clc_ctx* ctx, void* h_dst, int n, clc_desc_handle* handle /* nullable */);

/* The device-resident front end for the frames of n_images <= CLC_MAX_BATCH cameras at once (GPUDetector::detectAndDescribe,
 * GPUDetector.hpp:216-291, once per drone in ColoC::processImages, coloc.hpp:150-163): ONE pyramid launch, TWO detector launches
 * and -- when d_desc is not NULL -- ONE CLATCH launch for all of them, nothing synchronised, nothing copied back.  d_imgs[b]: u8
 * width x height device image (pitch bytes per row, the DetectorOptions size); d_kps[b]: room for DetectorOptions.maxkp keypoints;
 * d_counts[b]: uint32[2] on the device, {written, found}; d_desc[b]: maxkp x 64 B (rows past the count are left alone).  Same
 * keypoints, order and descriptors as n_images times clc_pyramid_build_dev + clc_detect_dev + clc_describe_detected_dev.  The
 * pointer arrays themselves are host memory.  Afterwards camera 0's pyramid is the context's current one. */
int clc_detect_batch_dev(clc_ctx* ctx, int n_images, const void* const* d_imgs, uint32_t width, uint32_t height, size_t pitch,
                         clc_keypoint* const* d_kps, uint32_t* const* d_counts, void* const* d_desc, void* stream);

/* ---- describe: replaces CLATCH() (CLATCH.h:168) + GPUDetector.hpp:280-290 ------------------ */

/* n keypoints on the current pyramid -> n x 64 B descriptors. */
int clc_describe(clc_ctx* ctx, const clc_keypoint* h_kps, int n, uint8_t* h_desc);
int clc_describe_dev(clc_ctx* ctx, const clc_keypoint* d_kps, int n, void* d_desc, void* stream);
/* A host that owns several cameras on one GPU (BASELINE configs 2-3: ColoC::processImages walks the drones one
 * by one, coloc.hpp:150-163) hands over the frames of all n_images <= CLC_MAX_BATCH cameras at once: ONE pyramid
 * launch and ONE CLATCH launch for all of them instead of one pair per camera -- same results as n_images times
 * clc_pyramid_build_dev + clc_describe_dev.  d_imgs[b]: u8 width x height device image (pitch bytes per row, the
 * DetectorOptions size); d_kps[b] / counts[b]: that camera's keypoints; d_desc[b]: counts[b] x 64 B out.  The
 * pointer arrays themselves are host memory.  Afterwards camera 0's pyramid is the context's current one. */
int clc_describe_batch_dev(clc_ctx* ctx, int n_images, const void* const* d_imgs, uint32_t width, uint32_t height,
                           size_t pitch, const clc_keypoint* const* d_kps, const int* counts, void* const* d_desc,
                           void* stream);
/* {scale*x, scale*y, 7*scale, angle}, scale = pow(1.2f, level): GPUDetector.hpp:172-179. */
int clc_keypoints_to_features(const clc_keypoint* h_kps, int n, float* h_feat4);

/* ---- match: replaces CUDAK2NN() (CUDAK2NN.h:54) + GPUMatcher.hpp:180-226 -------------------- */

/* For each query the best train index if (second_best - best > (uint8_t)threshold), else -1
 * (CUDAK2NN.cu:75; the kernel's threshold parameter is uint8_t, :46, so the int is truncated).
 * Ties for the minimum keep the lowest train index.  nt == 0 gives -1 everywhere (undefined in
 * the reference).  h_best / h_second (nullable) receive the two distances saturated to 65535
 * (the reference's 100000 / 200000 "none" sentinels read as 65535). */
int clc_match_2nn(clc_ctx* ctx, const void* h_q, int nq, const void* h_t, int nt, int threshold,
                  int32_t* h_match, uint16_t* h_best, uint16_t* h_second);
/* Detector -> matcher hand-over without a second upload.  The reference passes descriptors between the two through host memory
 * (FeatureMap regions, GPUDetector.hpp:181 -> GPUMatcher.hpp:188-196) and uploads them again in every match call.  Here the front end
 * keeps a frame's descriptors in a device block of a process-wide, per-device table; when the host stores the rows at a host address it
 * PUBLISHES that fact (clc_detect_store_descriptors does both; clc_desc_cache_publish for rows copied by the caller, d_src = where they
 * lie on the device, NULL = the frame this context staged last), and clc_match_2nn / clc_match_map / clc_match_pairs given that ADDRESS
 * and COUNT read the rows where they already are -- as far as the looking-up context's mode allows:
 *   CLC_DESC_CACHE_VERIFY (default, the policy classes included): optimistic and checked -- the sweep starts on the device rows at once,
 *       and while it runs the host folds ALL rows of the block it was handed (64-bit position-keyed fold) and compares with the fold
 *       taken when the block was published; a block edited anywhere since is uploaded and the sweep repeated.  Edited host rows are
 *       never matched stale; an unchanged block costs no upload and no extra latency (the fold hides behind the sweep);
 *   CLC_DESC_CACHE_TRUST: address, count, generation and 18 sampled rows; the caller STATES that it does not edit published blocks in
 *       place (opt-in: HIPMatcher::trustPublishedRegions(true));
 *   CLC_DESC_CACHE_OFF: every block is uploaded, like the reference (GPUMatcher.hpp:188-196); the front end publishes nothing.
 * A block published by a TRUST context carries no fold and is invisible to VERIFY lookups.  Up to 32 blocks, least recently used
 * replaced; CLC_DESC_CACHE=0|verify|trust in the environment sets the mode contexts start with. */
enum { CLC_DESC_CACHE_OFF = 0, CLC_DESC_CACHE_VERIFY = 1, CLC_DESC_CACHE_TRUST = 2 };
int clc_desc_cache_mode(clc_ctx* ctx, int mode);
int clc_desc_cache_publish(clc_ctx* ctx, const void* d_src, const void* h_desc, int n, clc_desc_handle* handle /* nullable */);
int clc_desc_handle_live(const clc_desc_handle* handle);
int clc_desc_cache_clear(void);
/* lookups of the host-pointer match entry points answered from the cache / uploaded, since the process started */
int clc_desc_cache_stats(unsigned long long* hits, unsigned long long* misses);
/* Device-resident form; d_q / d_t must be 16-byte aligned. */
int clc_match_2nn_dev(clc_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt,
                      int threshold, int32_t* d_match, void* stream);
/* Describe both cameras of a pair and match them as ONE step: replaces detectAndDescribe of each camera (GPUDetector.hpp:216-291)
 * followed by computeMatchesPair (GPUMatcher.hpp:165-172 -> :180-226).  Camera 0 is the query side (regions[pair.first]), camera 1 the
 * train side; d_desc[b] receives counts[b] x 64 B, d_match counts[0] indices into camera 1 (or -1), exactly what clc_describe_batch_dev +
 * clc_match_2nn_dev give: one pyramid launch, one CLATCH launch (train camera dispatched first), one sweep launch on the caller's stream.
 * What overlaps describe and sweep on MI355X is two contexts on two streams taking alternate steps (bench.py's headline loop), not a
 * split inside one step (profiles/r05_step_overlap.txt).  Enqueue only; capturable.  16-byte aligned descriptor pointers. */
int clc_describe_match_pair_dev(clc_ctx* ctx, const void* const* d_imgs, uint32_t width, uint32_t height, size_t pitch,
                                const clc_keypoint* const* d_kps, const int* counts, void* const* d_desc, int threshold,
                                int32_t* d_match, void* stream);
/* Many (query-slice, train-set) jobs over one descriptor arena in ONE sweep launch: the
 * all-pairs loop of GPUMatcher::computeMatches (GPUMatcher.hpp:143-155) and the per-rank share
 * of it after the multi-GPU all-gather.  h_jobs is host memory. */
int clc_match_jobs_dev(clc_ctx* ctx, const void* d_desc_base, const clc_match_job* h_jobs,
                       int njobs, int32_t* d_match, void* stream);
/* The same with the row counts of the two sets of every job living in DEVICE memory (the sweep reads them; no host
 * round trip between whatever produced the counts and this call): job j's nq / nt are the PLANNED sizes (grid and result
 * layout), its query slice starts at row q_row0[j] of a set with *d_cnt_q[j] valid rows, its train set has
 * min(nt, *d_cnt_t[j]) rows.  Planned query rows past the count are answered -1. */
int clc_match_jobs_counted_dev(clc_ctx* ctx, const void* d_desc_base, const clc_match_job* h_jobs, int njobs,
                               const int32_t* const* d_cnt_q, const int32_t* const* d_cnt_t, const uint32_t* q_row0,
                               int32_t* d_match, void* stream);

/* Which formulation of the all-pairs sweep a context uses (same results bit for bit, same workspace):
 * CLC_K2NN_MATRIX (default): bits as +-1 FP4 values on the matrix pipe, exact distances in the fp32 accumulator;
 * CLC_K2NN_POPCOUNT: xor + popcount on the vector ALU, the literal form of CUDAK2NN.cu:58-66 (kept for A/B timing).
 * Also settable at context creation through the environment, CLC_K2NN_FORMULATION=matrix|popcount. */
enum { CLC_K2NN_MATRIX = 0, CLC_K2NN_POPCOUNT = 1 };
int clc_k2nn_set_formulation(clc_ctx* ctx, int formulation);
/* Queries per sweep workgroup of the context's formulation: the grain on which a caller that deals query slices out
 * to several GPUs (clc_match_job.q_offset / nq) should cut them, so that no workgroup is split between two jobs. */
int clc_k2nn_queries_per_block(const clc_ctx* ctx);
/* How the context would cut ONE nq x nt pair into sweep workgroups (planning only, nothing is launched): info[0] query blocks, [1] train
 * splits per query block, [2] train rows per equal split, [3] 1 = splits folded in-launch (atomic top-2 rows), 0 = slabs + merge kernel,
 * [4], [5] train TILES (32 rows) of a split on wave slot 0 / 1 when the pair runs as one round with unequal shares by wave slot
 * (0 = equal shares; round 4, k2nn.hip), [6] queries per workgroup, [7] workgroups aimed at per launch.  The reference has no such
 * entry (its grid is ((num_q - 1) >> 8) + 1 blocks, CUDAK2NN.cu:79); tests and bench.py report the plan with it. */
int clc_k2nn_plan_query(const clc_ctx* ctx, int nq, int nt, int32_t* info);
/* What the sweep planner knows about the context's device and which unequal shares it uses (round 5): info[0] XCDs
 * (hipDeviceAttributeNumberOfXccs), [1] CUs, [2] the formulation's default workgroups per launch (3 per CU: one resident round of the
 * matrix sweep), [3], [4] share of a slot-0 / slot-1 workgroup in 1/256 of the equal share, [5] where that pair comes from: 0 built-in
 * default, 1 CLC_K2NN_BIAS, 2 a timed probe of four candidate pairs on this device (once per process and device, at the first matcher
 * context's creation; CLC_K2NN_PROBE=0 skips it), [6] XCDs the sweep kernel's workgroup map is compiled for, [7] CLC_K2NN_TARGET_BLOCKS
 * (0: unset).  probe_us (nullable, 4 floats): the 10k x 10k sweep's time under the candidates 295:264, 311:256, 326:249, 326:233. */
int clc_k2nn_device_info(const clc_ctx* ctx, int32_t* info, float* probe_us);

/* Measurement aid (replaces nothing in the reference): runs ONE sweep of d_q x d_t with the diagnostic build of the
 * matrix-formulation kernel, whose workgroups bracket their tile loop with the shader-clock and the constant 100 MHz
 * real-time counters, and returns the clock the chip actually held inside the kernel (median / min / max over the
 * workgroups, GHz).  Call it straight after a sustained run: peaks quoted at 2.4 GHz are only reached if this says so.
 * The results in d_match are the normal ones (threshold 40). */
int clc_k2nn_clock_check(clc_ctx* ctx, const void* d_q, int nq, const void* d_t, int nt, int32_t* d_match, void* stream,
                         double* ghz_median, double* ghz_min, double* ghz_max, int* workgroups);

/* The all-pairs loop of GPUMatcher::computeMatches (GPUMatcher.hpp:143-155) on host buffers: uploads
 * each camera's descriptors once (the reference re-uploads both sides for every pair,
 * GPUMatcher.hpp:188-196), sweeps every listed (first, second) pair in one launch group (Q = first,
 * T = second) and downloads one int32 array of counts[first] entries per pair into h_match[p]. */
int clc_match_pairs(clc_ctx* ctx, const void* const* h_desc, const int* counts, int ncams,
                    const int* pairs /* npairs x {first, second} */, int npairs, int threshold,
                    int32_t* const* h_match);

/* Map database: GPUMatcher::setMapData (GPUMatcher.hpp:110-117) / matchFeaturesWithMap (:252-271). */
int clc_set_map(clc_ctx* ctx, const void* h_desc, int n);
int clc_match_map(clc_ctx* ctx, const void* h_q, int nq, int threshold, int32_t* h_match);
/* The same against descriptors that are still on the GPU (clc_describe*_dev output): enqueue only; d_match[i] = map
 * index or -1.  In a streaming loop the frame's descriptors then never leave the device, only the nq x 4 B result does. */
int clc_match_map_dev(clc_ctx* ctx, const void* d_q, int nq, int threshold, int32_t* d_match, void* stream);

/* ---- pose scoring: the data-parallel core of SfM_Localizer::Localize (Localizer.hpp:82-93) --- */

/* err[h*N+i] = || x_i - hnormalized(K (R_h X_i + t_h)) ||^2, fp64.
 * h_Rt: H x 12 row-major [R|t]; h_X: N x 3; h_x: N x 2; h_K: 9 row-major. */
int clc_pnp_residuals(clc_ctx* ctx, const double* h_Rt, int H, const double* h_X,
                      const double* h_x, int N, const double* h_K, double* h_err);
/* Fused scoring without materialising the H x N matrix: per hypothesis the inlier count
 * (err < thr2) and the truncated cost sum_i min(err, thr2).  Outputs nullable. */
int clc_pnp_score(clc_ctx* ctx, const double* h_Rt, int H, const double* h_X, const double* h_x,
                  int N, const double* h_K, double thr2, int32_t* h_count, double* h_cost);

/* ---- two-view scoring: the data-parallel core of RobustMatcher::filterEssential -------------------
 * (RobustMatcher.hpp:153-186: AC-RANSAC over FivePointSolver + SymmetricEpipolarDistanceError.)
 * h_F: H x 9 row-major FUNDAMENTAL matrices (F = K2^-T E K1^-1); h_x1 / h_x2: N x 2 pixel points of the
 * two views.  err[h*N+i] = (x2^T F x1)^2 (1/|(F x1)_xy|^2 + 1/|(F^T x2)_xy|^2) / 4, fp64.  The minimal
 * solver stays with the caller (OpenMVG's, when linked in): generate all hypotheses, score once. */
int clc_epipolar_residuals(clc_ctx* ctx, const double* h_F, int H, const double* h_x1, const double* h_x2,
                           int N, double* h_err);
int clc_epipolar_score(clc_ctx* ctx, const double* h_F, int H, const double* h_x1, const double* h_x2, int N,
                       double thr2, int32_t* h_count, double* h_cost);

/* Essential-matrix RANSAC on the GPU -- the role of RobustMatcher::filterEssential (RobustMatcher.hpp:153-186:
 * ACRANSAC over essential::kernel::FivePointSolver + SymmetricEpipolarDistanceError): S minimal samples of 5
 * correspondences -> one five-point problem per lane (<= 10 essential matrices each, csrc/fivept.h) -> all
 * 10 S hypotheses scored over all N pixel correspondences (F = K2^-T E K1^-1) -> best = most inliers
 * (err < thr2), then lowest truncated cost, then lowest index -> inlier mask.  h_samples: S x 5 indices
 * (NULL = drawn from a xorshift64* stream seeded with `seed`).  Outputs (nullable): h_E, h_F 9 doubles
 * row-major, h_inlier_mask N bytes, *n_inliers (0 = no model).  OpenMVG's a-contrario threshold selection and
 * its solver's root order are unpinned (absent submodule). */
int clc_essential_ransac(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1,
                         const double* h_K2, const int32_t* h_samples, int S, uint64_t seed, double thr2,
                         double* h_E, double* h_F, uint8_t* h_inlier_mask, int* n_inliers);
/* The five-point hypotheses alone: h_E_out receives S x 10 x 9 doubles (NaN = no solution). */
int clc_essential_fivepoint(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1,
                            const double* h_K2, const int32_t* h_samples, int S, double* h_E_out);

/* Whole robust pose solve on the GPU -- the role of SfM_Localizer::Localize(P3P, max_iteration = 256)
 * at Localizer.hpp:82-93: S minimal samples -> one P3P problem per lane (<= 4 poses each) -> all
 * 4 S hypotheses scored over all N correspondences in one launch -> best = most inliers (err < thr2),
 * then lowest truncated cost, then lowest index -> its inlier mask.  h_samples: S x 3 point indices
 * (NULL = drawn from a xorshift64* stream seeded with `seed`; OpenMVG draws from std::mt19937 and
 * scores a contrario -- unpinned).  Outputs: h_Rt 12 doubles row-major [R|t],
 * h_inlier_mask N bytes (nullable), *n_inliers, *cost (nullable).  *n_inliers == 0 means no pose. */
int clc_pnp_ransac(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K,
                   const int32_t* h_samples, int S, uint64_t seed, double thr2, double* h_Rt,
                   uint8_t* h_inlier_mask, int* n_inliers, double* cost);
/* Single-pose refinement + covariance -- the role of Localizer::refine -> PoseRefiner::refinePose
 * (Localizer.hpp:110-177, Refiner.hpp:47-239; Ceres LM, HuberLoss(Square(4.0)), structure and intrinsics
 * fixed): Levenberg-Marquardt on 1/2 sum rho(||obs - proj||^2) over the 6 parameters [angle-axis | t],
 * from the initial pose h_Rt_in, using the correspondences with h_inlier_mask[i] != 0 (NULL = all).
 * Outputs (all nullable): h_Rt_out 12 doubles, h_cov 36 doubles = (J^T W J)^-1 in the [angle-axis | t]
 * parametrisation (row-major), *rmse = sqrt(final_cost / (2 n_used)) (Refiner.hpp:226), *iterations.
 * huber_a <= 0 selects the reference's 16.  Ceres is absent here, so results are defined by this
 * cost, not by Ceres' iterates (unpinned). */
int clc_pnp_refine(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K,
                   const uint8_t* h_inlier_mask, const double* h_Rt_in, double huber_a, int max_iter,
                   double* h_Rt_out, double* h_cov, double* rmse, int* iterations);

/* clc_pnp_ransac followed by clc_pnp_refine on its inliers as ONE submission (one upload, chained
 * kernels, one download): what Localizer::localizeImage does end to end (Localizer.hpp:77-108).
 * Outputs nullable; *n_inliers == 0 means no pose (h_Rt / h_cov are zero then). */
int clc_pnp_localize(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K,
                     const int32_t* h_samples, int S, uint64_t seed, double thr2, double huber_a,
                     double* h_Rt, double* h_cov, uint8_t* h_inlier_mask, int* n_inliers, double* rmse);

/* The hypotheses of the minimal solver alone: h_Rt_out receives 4 S x 12 doubles (NaN = no solution). */
int clc_pnp_p3p(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K,
                const int32_t* h_samples, int S, double* h_Rt_out);

/* ---- several cameras on several GPUs (no counterpart in the reference) ------------------------------------------------
 * The reference matches its cameras in a serial all-pairs loop inside one process (GPUMatcher::computeMatches,
 * GPUMatcher.hpp:143-155 over Utils::handlePairs, colocUtils.hpp:58-61).  Here: one process (and context) per GPU, rank r
 * owns camera r; clc_mc_gather_dev exchanges the descriptor blocks over xGMI, clc_mc_match_dev sweeps this rank's
 * contiguous share of the flattened (pair, query block) sequence; the union over the ranks is exactly the serial loop's
 * per-pair result (Q = descriptors of `first`, T = of `second`).  RCCL is loaded at run time (dlopen), only when
 * world > 1.  Exchange forms: CLC_MC_RCCL = ncclAllGather of the fixed-capacity blocks; CLC_MC_PEER_COPY = every rank
 * copies its block into each peer's arena through IPC-mapped pointers (one xGMI hop each) and a 4-byte all-gather of
 * the counts acts as the fence. */
#define CLC_MC_ID_BYTES 128
enum { CLC_MC_RCCL = 0, CLC_MC_PEER_COPY = 1 };
typedef struct clc_mc clc_mc;
/* one contiguous run of query rows of pair (first, second) assigned to a rank */
typedef struct clc_mc_share {
    int32_t  first, second;  /* camera ids, first < second                                  */
    uint32_t q_begin, nq;    /* query rows [q_begin, q_begin + nq) of camera `first`         */
    uint32_t out_offset;     /* first int32 of this run inside the rank-local result buffer */
} clc_mc_share;
/* Pure host arithmetic (no GPU, no context): the shares of `rank` when `world` ranks split the all-pairs work of
 * cameras with counts[c] descriptors, cut on `grain` queries (clc_k2nn_queries_per_block).  The C twin of
 * coloc_amd/multicam.py shard_pairs. */
int clc_mc_plan(const int* counts, int ncams, int world, int rank, int grain, clc_mc_share* out, int capacity, int* n_out);
/* Rank 0 creates the rendezvous id; the HOST application hands it to the other ranks (MPI, a socket, a file ...). */
int clc_mc_unique_id(uint8_t id[CLC_MC_ID_BYTES]);
/* Joins the `world`-rank group on ctx's device; maxkp = capacity of a camera's descriptor block (the same on all
 * ranks).  world == 1 needs no id and no RCCL (with an id it builds a one-rank communicator and every exchange goes
 * through the same RCCL / IPC calls as world > 1).  world > 1 with id == NULL creates a REHEARSAL handle without a
 * communicator: the calling process plays the other ranks with clc_mc_virtual_put (how the one-GPU tests drive every
 * rank's share through this entry); clc_mc_gather_dev then only files the rank's own block. */
int clc_mc_create(clc_ctx* ctx, const uint8_t id[CLC_MC_ID_BYTES], int world, int rank, int maxkp, clc_mc** out);
int clc_mc_virtual_put(clc_mc* mc, int other_rank, const void* d_desc, int count, void* stream);
/* Collective over the group (every rank calls it): exchanges the arenas' IPC handles through the communicator and maps the peers'
 * arenas, which the first CLC_MC_PEER_COPY exchange would otherwise do.  Calling it up front lets a host learn that the mapping is
 * not possible on this machine (status != CLC_OK) BEFORE any rank waits in an exchange. */
int clc_mc_open_peers(clc_mc* mc, void* stream);
int clc_mc_destroy(clc_mc* mc);
const char* clc_mc_last_error_string(const clc_mc* mc);
/* The arena [world][maxkp][64 B] the LAST exchange filled (one of the handle's two buffers: it alternates per exchange). */
int clc_mc_arena(const clc_mc* mc, void** d_arena, int* world, int* maxkp);
/* Exchange: this rank's descriptors (d_my_desc: a device buffer of maxkp rows, my_count of them valid) go to every
 * rank's arena, the counts come back to the host (h_counts_out: world ints, nullable).  Synchronises the stream. */
int clc_mc_gather_dev(clc_mc* mc, const void* d_my_desc, int my_count, int mode, int* h_counts_out, void* stream);
/* Sweep this rank's shares of the gathered arena (enqueue only): d_match receives the runs back to back
 * (share.out_offset), h_shares the shares themselves.  Needs the counts on the host: follows clc_mc_gather_dev. */
int clc_mc_match_dev(clc_mc* mc, int threshold, int32_t* d_match, int match_capacity, clc_mc_share* h_shares,
                     int share_capacity, int* n_shares, void* stream);
/* The same step WITHOUT a host synchronisation between the exchange and the sweep (a streaming host enqueues step after
 * step): clc_mc_gather_enqueue_dev only enqueues the exchange -- my_count from the host, or d_my_count != NULL: read on the
 * device (e.g. the detector's count, clc_detect_buffers), the whole capacity block then travels --, and
 * clc_mc_match_enqueue_dev cuts the shares on the block CAPACITY (identical for every step of a handle) and lets the sweep
 * read the gathered counts from device memory: planned query rows past a camera's count are answered -1, train rows past
 * it are not swept.  Per pair and valid row the result is the serial loop's.  clc_mc_counts synchronises the stream and
 * returns the counts of the last exchange.
 * Ordering contract (both gather forms): the arena is double-buffered by step parity, every rank calls the gather the same
 * number of times, and a rank enqueues step k + 1's exchange on the stream that holds its step-k sweep; then no peer copy
 * lands in a buffer a sweep still reads (coloc_amd/csrc/multicam.hip, top). */
int clc_mc_gather_enqueue_dev(clc_mc* mc, const void* d_my_desc, int my_count, const int32_t* d_my_count, int mode, void* stream);
int clc_mc_match_enqueue_dev(clc_mc* mc, int threshold, int32_t* d_match, int match_capacity, clc_mc_share* h_shares,
                             int share_capacity, int* n_shares, void* stream);
int clc_mc_counts(clc_mc* mc, int* h_counts_out, void* stream);
/* OVERLAPPED steps (round 6; default off, call before the handle's first exchange and before clc_mc_open_peers): the caller passes one
 * stream to clc_mc_gather_enqueue_dev -- and enqueues its describe on it -- and ANOTHER to clc_mc_match_enqueue_dev; step k + 1's
 * describe + exchange then run beside step k's sweep.  The handle keeps three arena buffers and orders the two streams with events
 * (sweep k behind exchange k; exchange k behind sweep k - 2); all collectives stay on the exchange stream, so one communicator serves.
 * Results are those of the one-stream step.  The context's front-end work (pyramid, CLATCH) must be enqueued on the exchange stream and
 * nothing but the sweep on the other one: a context's matcher workspace and its detector workspace are separate. */
int clc_mc_set_overlap(clc_mc* mc, int on);
/* What RCCL itself says about the communicator behind the handle: *n_ranks = ncclCommCount, *user_rank = ncclCommUserRank
 * (*n_ranks == 0: the handle has no communicator -- one rank without an id, or a rehearsal handle). */
int clc_mc_comm_info(const clc_mc* mc, int* n_ranks, int* user_rank);

/* ---- a-contrario model selection: what the reference actually runs ------------------------------------------------------
 * Localizer::localizeImage calls SfM_Localizer::Localize(P3P_KE_CVPR17, ..., {error_max = +inf, max_iteration = 256})
 * (include/coloc/Localizer.hpp:82-93) and RobustMatcher::filterEssential calls robust::ACRANSAC over the five-point
 * kernel with precision +inf (include/coloc/RobustMatcher.hpp:153-171): OpenMVG's AC-RANSAC, which needs NO threshold --
 * per model it sorts the residuals, evaluates the number of false alarms NFA(k) of "the k best are inliers" for every k
 * and keeps the model / k of lowest NFA; after the first meaningful model (NFA < 0) the reserved 10 % of the iterations
 * sample among its inliers.  (Moisan, Moulon, Monasse, IPOL 2012; OpenMVG itself is an empty submodule in the reference
 * tree, so its RNG stream and solver root order are unpinned: samples here are a documented pure function of
 * (seed, iteration, index set), coloc_amd/csrc/clc_acr.h.)  Batches of iterations are evaluated per round on the GPU --
 * minimal solves, one sort + NFA scan per model in LDS, a sequential-semantics selection -- with results identical to
 * the iteration-by-iteration loop (oracle/clc_oracle_acr.c).  precision = INFINITY is the reference's setting; a finite
 * value is OpenMVG's upper bound on the inlier residual, in pixels^2.  At most 16384 correspondences per solve (CLC_ERR_CAPACITY beyond: the per-model sort runs in one workgroup's LDS).
 *
 * clc_pnp_acransac: h_X N x 3, h_x N x 2 UNDISTORTED pixels, h_K 9 row-major (fx = K[0] scales residuals to the
 * normalised camera plane as ACKernelAdaptorResection_Intrinsics does).  Outputs (nullable): h_Rt 12 doubles [R|t],
 * h_inlier_mask N bytes, h_inliers the inlier indices in ascending residual order (capacity N; vec_inliers),
 * *n_inliers (0 = no meaningful model; the caller applies Localize's "> 2.5 x 3" test), *error_max the precision found
 * in pixels (ACRansacOut.first), *min_nfa the log10 NFA (ACRansacOut.second), *iterations actually run. */
int clc_pnp_acransac(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, int max_iteration,
                     uint64_t seed, double precision, double* h_Rt, uint8_t* h_inlier_mask, int32_t* h_inliers,
                     int* n_inliers, double* error_max, double* min_nfa, int* iterations);
/* The same followed by clc_pnp_refine on the inliers in one submission: Localizer::localizeImage end to end
 * (Localizer.hpp:77-108).  h_cov 36 doubles, *rmse as clc_pnp_refine. */
int clc_pnp_localize_ac(clc_ctx* ctx, const double* h_X, const double* h_x, int N, const double* h_K, int max_iteration,
                        uint64_t seed, double precision, double huber_a, double* h_Rt, double* h_cov,
                        uint8_t* h_inlier_mask, int32_t* h_inliers, int* n_inliers, double* error_max, double* rmse);
/* Several independent localisations at once -- BASELINE config[2]'s "batched PnP/RANSAC pose", one per camera: job i is what
 * clc_pnp_acransac (refine == 0) or clc_pnp_localize_ac (refine != 0) does, on ctxs[i] (every job needs a context of its own; contexts created
 * without detector and matcher options are enough).  A solve is a chain of short launches with the host in the loop and leaves the GPU idle
 * most of the time; here ONE host thread drives all the chains, so they interleave on the device.  Every job's result is the one the
 * single-solve entry gives for the same arguments (same model, inliers, threshold, covariance, bit for bit).  jobs[i].status holds the
 * job's own status; the return value is the first failure, CLC_OK if none.
 * Since round 5 the solves of a larger batch (eight or more here, four or more two-view filters below) SHARE their launches instead --
 * round r of every unfinished solve is one launch on ctxs[0]'s stream, blockIdx.y = solve -- which changes the schedule, not a result;
 * on return the streams of all the contexts are ordered behind whatever of those launches is still in flight (CLC_ACR_LOCKSTEP=0|1
 * forces the form). */
typedef struct clc_pose_job {
    /* in */
    const double* X;          /* n x 3 world points                                      */
    const double* x;          /* n x 2 pixels                                            */
    const double* K;          /* 3 x 3 row-major                                         */
    int           n;
    int           max_iteration;
    uint64_t      seed;
    double        precision;  /* +inf: a-contrario threshold (Localizer.hpp:82-84)       */
    int           refine;     /* != 0: LM refinement + 6 x 6 covariance behind the solve */
    double        huber_a;    /* <= 0: 16                                                */
    /* out (pointers nullable) */
    double*       Rt;         /* 12                                                      */
    double*       cov;        /* 36 (refine)                                             */
    uint8_t*      inlier_mask;/* n                                                       */
    int32_t*      inliers;    /* n                                                       */
    int           n_inliers;
    int           iterations;
    int           status;
    double        error_max;
    double        rmse;       /* refine                                                  */
} clc_pose_job;
int clc_pnp_localize_ac_batch(clc_ctx* const* ctxs, clc_pose_job* jobs, int n_jobs);

/* RobustMatcher::filterEssential's ACRANSAC: h_x1 / h_x2 N x 2 undistorted pixels, K1 / K2, img_w x img_h the size of
 * image 2 (point-to-line alpha0 = 2 D / A / 2, residual^(1/2)); residual = symmetric epipolar distance of
 * F = K2^-T E K1^-1 as clc_epipolar_residuals.  *error_max is that squared distance at the a-contrario threshold. */
int clc_essential_acransac(clc_ctx* ctx, const double* h_x1, const double* h_x2, int N, const double* h_K1,
                           const double* h_K2, int img_w, int img_h, int max_iteration, uint64_t seed, double precision,
                           double* h_E, double* h_F, uint8_t* h_inlier_mask, int32_t* h_inliers, int* n_inliers,
                           double* error_max, double* min_nfa, int* iterations);

/* Several two-view filters at once (round 5): what RobustMatcher::filterMatches runs pair after pair (RobustMatcher.hpp:455-483 ->
 * filterEssential :153-171).  As clc_pnp_localize_ac_batch: one context per job (all on one device), ONE host thread drives all the solves
 * (chains of launches of their own that interleave on the device, or -- four or more jobs -- rounds in launches shared by the batch);
 * every job's result is clc_essential_acransac's for the same arguments. */
typedef struct clc_two_view_job {
    /* in */
    const double* x1;         /* n x 2 undistorted pixels, image 1                        */
    const double* x2;         /* n x 2, image 2                                           */
    const double* K1;         /* 3 x 3 row-major                                          */
    const double* K2;
    int           n, img_w, img_h, max_iteration;
    uint64_t      seed;
    double        precision;  /* +inf: a-contrario threshold                              */
    /* out (pointers nullable) */
    double*       E;          /* 9                                                        */
    double*       F;          /* 9                                                        */
    uint8_t*      inlier_mask;/* n                                                        */
    int32_t*      inliers;    /* n                                                        */
    int           n_inliers, iterations, status;
    double        error_max, min_nfa;
} clc_two_view_job;
int clc_essential_acransac_batch(clc_ctx* const* ctxs, clc_two_view_job* jobs, int n_jobs);

/* The same a-contrario filter under RobustMatcher's other two models (ABI 4; RobustMatcher.hpp:399-405 dispatches on colocParams::model):
 *   CLC_MODEL_FUNDAMENTAL 'F'  filterFundamental :128-151  ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError, UnnormalizerT>(.., true)
 *   CLC_MODEL_HOMOGRAPHY  'H'  filterHomography  :188-239  ACKernelAdaptor<FourPointSolver, AsymmetricError, UnnormalizerI>(.., false)
 *   CLC_MODEL_ESSENTIAL   'E'  = clc_essential_acransac (h_M = E, h_F = F)
 * h_x1 / h_x2: N x 2 pixels; both images img_w x img_h (the reference hands params.imageSize for both): the points are conditioned by the
 * image size, samples of 7 (4) correspondences give <= 3 (1) models, residual = squared distance to the epipolar line in image 2
 * ('F', point-to-line alpha0, residual^(1/2)) / squared transfer error ('H', point-to-point alpha0), both in conditioned coordinates.
 * h_M (9, nullable): the model brought back to PIXELS (x2^T F x1 = 0; x2 ~ H x1) -- what RelativePose_Info::essential_matrix receives;
 * h_F (9, nullable): F for 'E' and 'F', zeros for 'H'.  *error_max: the a-contrario threshold in pixels (sqrt(e) / N2(0,0)); K1 / K2
 * are read for 'E' only.  The rounds are the resection's (one launch per round: replay + seven-/four-point solve + residuals, sort, NFA). */
enum { CLC_MODEL_ESSENTIAL = 'E', CLC_MODEL_FUNDAMENTAL = 'F', CLC_MODEL_HOMOGRAPHY = 'H' };
int clc_two_view_acransac(clc_ctx* ctx, int model, const double* h_x1, const double* h_x2, int N, const double* h_K1,
                          const double* h_K2, int img_w, int img_h, int max_iteration, uint64_t seed, double precision,
                          double* h_M, double* h_F, uint8_t* h_inlier_mask, int32_t* h_inliers, int* n_inliers,
                          double* error_max, double* min_nfa, int* iterations);
/* Several filters of ONE model at once, as clc_essential_acransac_batch (which is this with 'E'): job.E receives the model matrix (E, F
 * or H), job.F the fundamental matrix ('E', 'F') or zeros ('H'); job.K1 / K2 are read for 'E' only. */
int clc_two_view_acransac_batch(clc_ctx* const* ctxs, int model, clc_two_view_job* jobs, int n_jobs);
/* The minimal solver of 'F' / 'H' on caller-chosen samples (the hypothesis generator the tests hand to the sequential oracle, as
 * clc_pnp_p3p / clc_essential_fivepoint are for the other kinds): h_samples S x 7 (4) indices into the N correspondences; h_models
 * S x 3 (1) x 9 matrices IN CONDITIONED COORDINATES, NaN-filled where a sample has fewer real roots. */
int clc_two_view_minimal(clc_ctx* ctx, int model, const double* h_x1, const double* h_x2, int N, int img_w, int img_h,
                         const int32_t* h_samples, int S, double* h_models);

/* The inter-camera step of ColoC::interPoseEstimator(source, dest) between the pair's putative matches and the covariance intersection
 * (coloc.hpp:296-340), for several camera pairs at once: a-contrario five-point filter (filterMatchesPair, :296) -> relative pose from
 * E with the chirality vote (RobustMatcher.hpp:176-183) -> the pair's temporary map triangulated in the source camera's frame (:306) ->
 * the features it shares with the global map -> its scale against the global map through them (colocUtils.hpp:184-211) -> the
 * destination's pose through the source's, refined against the temporary map with its 6 x 6 covariance (refinePose, :340;
 * Huber(huber_a)).  tv.x1 = the SOURCE frame's features, tv.x2 = the destination's; tv.E and tv.inliers must be given.
 * The common features come, per job, from one of two places:
 *   THE REFERENCE'S CHAIN (d_first_desc, first_feature and d_map_desc given): setupMapDatabase(inter) keeps, for every point of the
 *       temporary map, the descriptor of its FIRST observation -- the pair's camera with the lower id (colocData.hpp:109-117) --, and
 *       matchMapFeatures(mapRegions, interMapRegions) matches the global map's descriptors against them: K2NN, Q = global map, T =
 *       temporary map, threshold 60 (coloc.hpp:317-323, GPUMatcher.hpp:157-163); commonFeatures = the accepted map points in ascending
 *       order.  Here: the rows are gathered on the device from that camera's descriptor block and swept against the map's block, which
 *       both stay where they are (device pointers);
 *   THE SHORTCUT (map_index given, the chain's pointers NULL): correspondence i's SOURCE feature is global map point map_index[i] (-1:
 *       none) -- what the source frame's own map tracking (matchSceneWithMap) already found; no descriptor work, common features in
 *       correspondence order.
 * stage says how far a job got. */
enum { CLC_INTER_OK = 0, CLC_INTER_NO_MODEL = 1 /* filter found < 13 inliers */, CLC_INTER_NO_RELATIVE_POSE = 2 /* < 8 points in front of both cameras */,
       CLC_INTER_NO_SCALE = 3 /* < 8 features shared with the global map */, CLC_INTER_NO_REFINEMENT = 4 };
typedef struct clc_inter_pose_job {
    clc_two_view_job tv;
    /* in */
    const int32_t* map_index; /* the shortcut: tv.n, index of correspondence i's SOURCE feature in the global map, -1 = none (nullable) */
    const double*  map_X;     /* global map points, 3 doubles each                         */
    int            map_n;     /* number of map points: an index outside [0, map_n) counts as "not a map feature" */
    const double*  Rt_source; /* 12: [R|t] of the source camera, x_cam = R X + t           */
    double         huber_a;   /* <= 0: 16                                                  */
    /* the reference's chain (all three, or none): */
    const void*    d_first_desc;   /* DEVICE: the descriptor block (rows of 64 B, 16-byte aligned) of the pair's camera with the lower id */
    const int32_t* first_feature;  /* tv.n: correspondence i's feature (row) in that block  */
    const void*    d_map_desc;     /* DEVICE: the global map's descriptors, map_n rows in map_X's order, 16-byte aligned */
    int            match_threshold;/* <= 0: 60 (GPUMatcher.hpp:162)                         */
    /* out */
    double         Rt[12];    /* the destination's pose through the source, refined        */
    double         cov[36];   /* [angle-axis | translation] order, as clc_pnp_refine       */
    double         rmse, scale;
    int            n_front, n_common, n_refined, stage;
    int            n_map_matches; /* the reference's chain: global map points matched to the temporary map (before the depth-ratio screen) */
} clc_inter_pose_job;
int clc_inter_pose_batch(clc_ctx* const* ctxs, clc_inter_pose_job* jobs, int n_jobs);

/* ---- fusion (host arithmetic; no GPU work) ---------------------------------------------------------
 * Covariance intersection of two 3-D position estimates as CoLoC fuses intra- and inter-camera poses
 * (include/coloc/CovIntersection.hpp:24-49, called at include/coloc/coloc.hpp:362-389): omega in [0,1]
 * minimising trace(inv(inv(CA) + inv(CB) - inv(omega CA + (1-omega) CB))) to 1e-3, then the fused
 * covariance (9, row-major) and position (3).  Same code as coloc_amd/host/HIPCovIntersection.hpp, exported
 * for hosts that bind the C ABI only.  ctx may be NULL. */
int clc_cov_intersection(const double* CA, const double* CB, const double* ca, const double* cb,
                         double* omega, double* cov_fused, double* pos_fused);

#ifdef __cplusplus
}
#endif
#endif /* COLOC_HIP_H */
