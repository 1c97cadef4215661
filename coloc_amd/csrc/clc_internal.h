// clc_internal.h -- shared declarations of libcoloc_hip.so's translation units (not installed).
#ifndef CLC_INTERNAL_H
#define CLC_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/coloc_hip.h"

namespace clc {

// ---- optional per-kernel event timing (clc_profile_* in the C ABI) ---------------------------
struct Profiler;
// Records an event on `stream` tagged (kernel, begin/end); no-op when prof is null or disabled.
void prof_mark(Profiler* prof, int kernel, bool begin, hipStream_t stream);

// ---- pyramid -------------------------------------------------------------------------------
struct LevelDesc {
    uint32_t w, h;        // level size
    uint32_t pitch;       // bytes per row (multiple of 64)
    uint32_t offset;      // byte offset of the level inside the pyramid arena
};
struct PyramidDesc {
    LevelDesc lv[CLC_MAX_LEVELS];
    float     f[CLC_MAX_LEVELS];      // resampling factor of level i (f_0 = 1)
    uint32_t  blk_begin[CLC_MAX_LEVELS + 1]; // first workgroup of level i in the fused resample launch
    int       levels;
};

// One launch: copy the caller's device image (d_src, src_pitch bytes per row) into level 0 of the
// arena and resample levels 1..L-1 from it.
hipError_t launch_pyramid(const PyramidDesc& pd, uint8_t* arena, const uint8_t* d_src, uint32_t src_pitch,
                          hipStream_t stream, Profiler* prof = nullptr);
// The same for the frames of n_img cameras in ONE launch: image b goes to the pyramid at arena + b * slot_stride.
static constexpr int kMaxBatch = CLC_MAX_BATCH;
hipError_t launch_pyramid_batch(const PyramidDesc& pd, uint8_t* arena, size_t slot_stride, const uint8_t* const* d_src,
                                int n_img, uint32_t src_pitch, hipStream_t stream, Profiler* prof = nullptr);

// ---- CLATCH ----------------------------------------------------------------------------------
hipError_t launch_clatch(const PyramidDesc& pd, const uint8_t* arena, const clc_keypoint* d_kps,
                         int n, uint64_t* d_desc, hipStream_t stream, Profiler* prof = nullptr);
// Keypoint lists of n_img cameras in ONE launch (blockIdx.y = camera; pyramid b at arena + b * slot_stride).
struct ClatchBatch {
    const clc_keypoint* kps[kMaxBatch];
    uint64_t*           desc[kMaxBatch];
    int                 n[kMaxBatch];
};
hipError_t launch_clatch_batch(const PyramidDesc& pd, const uint8_t* arena, size_t slot_stride, const ClatchBatch& batch,
                               int n_img, hipStream_t stream, Profiler* prof = nullptr);

// ---- detector (FAST-9 + NMS + orientation) ---------------------------------------------------
// Two launches for the pyramids of n_img cameras (pyramid b at arena + b * slot_stride, score map b at score + b * slot_stride):
// d_mask: n_img x detect_total_tiles() x 16 keypoint-mask words, d_tcount: n_img x detect_total_tiles() tile counts (both rewritten
// by every call: nothing to clear); d_count[b][0] = keypoints written to d_kps[b] (<= maxkp), d_count[b][1] = found.
hipError_t launch_detect(const PyramidDesc& pd, const uint8_t* arena, size_t slot_stride, int n_img, uint8_t* score, uint64_t* d_mask,
                         uint32_t* d_tcount, uint32_t threshold, uint32_t maxkp, clc_keypoint* const* d_kps, uint32_t* const* d_count,
                         hipStream_t stream, Profiler* prof = nullptr);
uint32_t detect_total_tiles(const PyramidDesc& pd);
// CLATCH with the keypoint count read from device memory (no host round trip after detect)
hipError_t launch_clatch_counted(const PyramidDesc& pd, const uint8_t* arena, const clc_keypoint* d_kps,
                                 const uint32_t* d_count, int max_n, uint64_t* d_desc, hipStream_t stream,
                                 Profiler* prof = nullptr);
// the same for n_img cameras in one launch: batch.n[b] = capacity of camera b's lists, d_count[b] = its count on the device
hipError_t launch_clatch_counted_batch(const PyramidDesc& pd, const uint8_t* arena, size_t slot_stride, const ClatchBatch& batch,
                                       const uint32_t* const* d_count, int n_img, hipStream_t stream, Profiler* prof = nullptr);

// ---- K2NN ------------------------------------------------------------------------------------
struct K2nnJobDev {
    const uint4* q;        // first query row of this job
    const uint4* t;        // first train row
    int32_t*     out;      // match indices, nq entries
    uint16_t*    best_out; // nullable
    uint16_t*    second_out; // nullable
    uint32_t     nq, nt;
    uint32_t     thr;      // already truncated to 8 bits
    uint32_t     qblocks;  // ceil(nq / queries-per-workgroup)
    uint32_t     splits;   // train-dimension splits
    uint32_t     t_per_split;
    uint32_t     partial_off; // first uint2 of this job's partial slab (slab mode) / of its top-2 row (atomic mode)
    uint32_t     nq_pad;      // row length of the slab
    uint32_t     atomic_merge;// 1: splits are folded with two atomicMin per query into a top-2 row, no slab
    uint32_t     cnt_off;     // atomic mode: first arrival counter (one per query block) of this job, in uint32 units
    // device-resident row counts (nullable; the multi-camera step plans its shares on the block CAPACITY and lets the sweep read the
    // gathered counts itself, so that no host synchronisation sits between the exchange and the sweep): with cnt_q the job covers
    // query rows [q_row0, q_row0 + nq) of a set of *cnt_q valid rows -- rows past the end are answered -1 --, with cnt_t the train
    // set has min(nt, *cnt_t) rows.  nq / nt stay the planned sizes (grid, splits, result layout).
    const int32_t* cnt_q;
    const int32_t* cnt_t;
    uint32_t     q_row0;
    uint32_t     xcd_rot;     // matrix sweep: the XCD (workgroup id & 7) that takes this job's query block 0 -- the jobs of a launch
                              // continue round the XCDs where the job before them stopped (k2nn_plan)
    // matrix sweep, ONE job that fills the chip once (round 4): unequal train shares by the wave slot a workgroup lands on.  The first
    // 256 workgroup ids take slot 0 of every SIMD, the next 256 slot 1, the rest slot 2 (measured: slot == id / 256 for 760 of 760), and the
    // matrix pipe serves a lower slot first: with equal shares the slot-0 workgroup of a CU ends its loop at 17.3 us, slot 1 at 19.0, slot 2
    // at 22.5, the last part of it alone on the CU at a quarter of the CU's rate.  bias_a / bias_b = train tiles of a slot-0 / slot-1 split
    // (0 = equal shares); the slot-2 splits of a query block take what is left.
    uint32_t     bias_a, bias_b;
    uint32_t     bias_magic;  // ceil(2^32 / qblocks): workgroup id / qblocks as a multiply-high (exact for ids below 2^16)
    uint32_t     slot_wgs;    // biased plans: workgroup ids per wave slot in the plan's id space (CUs of one XCD, or of the device)
};
// what the planner needs to know about the device (clc_ctx_create fills it from the HIP device attributes)
static constexpr uint32_t kK2nnXcds = 8;       // XCDs the sweep kernel's workgroup map is compiled for (every gfx950 part has eight)
struct K2nnDevice {
    uint32_t n_xcd = kK2nnXcds;                // hipDeviceAttributeNumberOfXccs
    uint32_t n_cu = 256;                       // multiProcessorCount
};
// the sizes a sweep workgroup works with (scalar loads when the counts live in device memory)
__device__ __forceinline__ uint32_t k2nn_job_nq(const K2nnJobDev& job)
{
    if (!job.cnt_q) return job.nq;
    const int left = *job.cnt_q - (int)job.q_row0;
    return left <= 0 ? 0u : ((uint32_t)left < job.nq ? (uint32_t)left : job.nq);
}
__device__ __forceinline__ uint32_t k2nn_job_nt(const K2nnJobDev& job)
{
    if (!job.cnt_t) return job.nt;
    const int c = *job.cnt_t;
    return c <= 0 ? 0u : ((uint32_t)c < job.nt ? (uint32_t)c : job.nt);
}
static constexpr int kK2nnJobsPerLaunch = 16;
// Passed BY VALUE as the kernel argument (1.2 KB of kernarg): no job upload, no staging hazard.
struct K2nnJobList {
    K2nnJobDev j[kK2nnJobsPerLaunch];
    // job 0 with unequal shares by wave slot (bias_a != 0), filled by launch_k2nn (no integer division in the kernel: ~30 vector
    // instructions each).  bias_magic == 0: what workgroup w = id >> 3 of an XCD takes -- query block of the XCD (bits 0..4), first train
    // tile (5..22), train tiles (23..31: k2nn.hip kBias*); bias_magic != 0: per query block, see launch_k2nn
    uint32_t bias_tab[128];
};
struct K2nnPlan {
    size_t   partial_elems;// uint2 entries needed
    bool     atomic_merge; // every job fits the 22-bit global train index -> atomic top-2 merge
};
// The two formulations of the sweep (k2nn.hip): FP4 matrix pipe (default) and round 1's xor + popcount VALU kernel.
enum { K2NN_MATRIX = 0, K2NN_POPCOUNT = 1 };
// Fill the derived fields of jobs[] (qblocks/splits/t_per_split/partial_off/nq_pad).
// bias_a / bias_b: train share of a workgroup on wave slot 0 / 1 in 1/256 of the equal share (0: equal shares); used by single-job one-round plans only
K2nnPlan k2nn_plan(K2nnJobDev* jobs, int njobs, int target_blocks, bool xcd_map = true, int formulation = K2NN_MATRIX, int bias_a = 0, int bias_b = 0,
                   K2nnDevice dev = K2nnDevice{});
// Sweep + merge over all jobs (chunks of kK2nnJobsPerLaunch per launch pair).
// In atomic mode d_partial must hold 0xFF bytes in every entry the jobs use (top-2 rows and arrival counters);
// the workgroup that completes a query block leaves it that way again (self re-arming workspace).
// d_stamps (matrix formulation only, nullable): run the diagnostic build whose workgroups record their shader-clock /
// real-time stamps, 4 uint64 per workgroup of the launch grid
// probe: the share probe's launches (the default matrix sweep under a symbol of its own, so that kernel traces keep them apart)
hipError_t launch_k2nn(const K2nnJobDev* jobs, int njobs, uint2* d_partial, hipStream_t stream,
                       Profiler* prof = nullptr, int formulation = K2NN_MATRIX, uint64_t* d_stamps = nullptr, bool probe = false);
int k2nn_queries_per_block(int formulation);

// ---- PnP -------------------------------------------------------------------------------------
hipError_t launch_pnp_residuals(const double* d_Rt, int H, const double* d_X, const double* d_x,
                                int N, const double* d_K, double* d_err, hipStream_t stream,
                                Profiler* prof = nullptr);
hipError_t launch_pnp_score(const double* d_Rt, int H, const double* d_X, const double* d_x, int N,
                            const double* d_K, double thr2, int32_t* d_count, double* d_cost,
                            hipStream_t stream, Profiler* prof = nullptr);

// P3P hypotheses for S minimal samples (4 slots each) -> score -> select + inlier mask of the winner.
// d_result receives one packed record {double Rt[12]; double cost; int32 h; int32 count} (pnp_result_bytes()).
// hs (nullable): the inputs are still in the caller's pinned host buffer [X 3N | x 2N | K 16 | samples]; the first
// launch reads them from there and stages them into d_X.. itself, the last one mirrors record + mask into pinned host
// memory -- no copy commands around the launches.
struct PnpHostStage {
    const double* src;      // pinned host input block
    int           n_doubles;// its length
    uint8_t*      h_mask;   // pinned host: N bytes
    void*         h_result; // pinned host: PnpResult
};
hipError_t launch_pnp_ransac(const double* d_X, const double* d_x, int N, const double* d_K, const int32_t* d_samples,
                             int S, double thr2, double* d_Rt, int32_t* d_count, double* d_cost, uint8_t* d_mask,
                             void* d_result, hipStream_t stream, Profiler* prof = nullptr, const PnpHostStage* hs = nullptr);
size_t pnp_result_bytes();
// five-point problems for S minimal samples (10 slots of {F(9), E(9)} each) -> symmetric-epipolar score -> select + mask.
// d_result: {double E[9]; double F[9]; double cost; int32 h; int32 count} (epi_result_bytes()).
hipError_t launch_essential_ransac(const double* d_x1, const double* d_x2, int N, const double* d_K1, const double* d_K2,
                                   const int32_t* d_samples, int S, double thr2, double* d_FE, int32_t* d_count, double* d_cost,
                                   uint8_t* d_mask, void* d_result, hipStream_t stream, Profiler* prof = nullptr);
size_t epi_result_bytes();
// symmetric epipolar distance of H fundamental matrices: d_err != null -> H x N residuals, else counts / costs
hipError_t launch_epipolar(const double* d_F, int H, const double* d_x1, const double* d_x2, int N, double thr2, double* d_err,
                           int32_t* d_count, double* d_cost, hipStream_t stream, Profiler* prof = nullptr);
// Levenberg-Marquardt refinement of one pose over the (masked) correspondences + 6x6 covariance.
// d_out: {double Rt[12]; double cov[36]; double cost; double rmse; int32 iterations; int32 n_used}
hipError_t launch_pnp_refine(const double* d_Rt_in, const double* d_X, const double* d_x, const uint8_t* d_mask, int N,
                             const double* d_K, double huber_a, int max_iter, void* d_out, hipStream_t stream,
                             Profiler* prof = nullptr, const int32_t* d_valid = nullptr, void* h_out = nullptr);
// (h_out: pinned host record written INSTEAD of d_out, so that no device-to-host copy command is needed)
size_t pnp_result_valid_offset();   // byte offset of the int32 "winning hypothesis" (< 0: none) in the ransac result record
size_t pnp_refine_out_bytes();
size_t pnp_refine_ready_offset();   // int32 written last by the refinement launch (1), for a host polling a pinned record

// ---- a-contrario RANSAC (acransac.hip) -----------------------------------------------------------------------
static constexpr int kAcrMaxBatch = 128;           // iterations evaluated per round
static constexpr int kAcrMaxN = 16384;             // correspondences per solve: 8 B x 16 384 = 128 KB of LDS per model slot (16 elements per thread)
static constexpr size_t kAcrMaxLds = (size_t)kAcrMaxN * 8;
struct AcrProblem {        // passed by value to every kernel of a solve
    int kind;              // 0: resection (P3P, [R|t] models of 12 doubles), 1: essential (five-point, {F, E} models of 18),
                           // 2: fundamental (seven-point), 3: homography (four-point): 9 doubles in a stride of 12, normalised coordinates
    int n, m, max_models, model_doubles;
    int batch_cap;         // most iterations a round evaluates (<= kAcrMaxBatch): the schedule, not the result, depends on it
    const double* a;       // X (3 n)  | x1 (2 n; kinds 2, 3: conditioned by the image size)
    const double* b;       // x (2 n)  | x2 (2 n)
    const double* K1;      // 9 doubles row-major (+ padding)
    const double* K2;
    const float* logc_n;   // log10 C(n, k), k = 0..n
    const float* logc_k;   // log10 C(k, m)
    double loge0, logalpha0, mult, max_threshold;
    double norm;           // resection: 1 / focal (residuals are scaled to the normalised camera plane); essential: 1; kinds 2, 3: N2(0,0)
    uint64_t seed;
    double K1v[9];         // resection: K1 by value -- kernel arguments sit in scalar registers, a load of K1 is a ~1 us round trip
};
struct AcrState {          // device resident; the host sees one packed 8-byte word of it after every round
    double min_nfa, error_max;
    double model[18];
    int32_t n_inliers, best_iter;
    int32_t iter, n_iter, reserve;
    int32_t n_index, index_all, ac_mode;
    int32_t rounds, last_batch;
    int32_t cur_batch;     // iterations the NEXT round evaluates (the solve / nfa kernels read it from here: rounds are enqueued
    int32_t grow;          // one ahead of the host's knowledge); grow = batch size while no event has happened (32, 64, 128)
    int32_t rounds_eval;   // rounds that evaluated at least one iteration (`rounds` also counts the empty round enqueued ahead)
    int32_t evaluated;     // acr_round_kernel: the slots of the other parity hold a batch of cur_batch iterations to replay (0: a fresh run)
};
struct AcrResult {
    double model[18];
    double min_nfa;
    double error_max;      // un-normalised: pixels (resection), squared pixels (essential)
    int32_t n_inliers;
    int32_t valid;         // iteration that produced the model, -1: no meaningful model
    int32_t iterations, rounds;
};
struct AcrHyp;
size_t acr_hyp_bytes();
// where the round that COMPLETES a run leaves the result (the finish work rides in that round's launch, and the host returns as soon as
// the polled word says "done" instead of launching a finish kernel behind the round enqueued ahead)
struct AcrFinish {
    uint8_t* d_mask; AcrResult* d_res;                 // device copies (the refinement reads them)
    uint8_t* h_mask; int32_t* h_inliers; AcrResult* h_res;     // pinned host memory (nullable)
};
// Everything the launches of ONE solve work on; up to kMaxBatch solves of one kind share a launch (blockIdx.y = chain): their rounds then
// advance in lockstep, one launch (resection) or two (two-view) per round for all of them (round 5: the batched entries were bound by the
// host's launch calls -- 8 poses = ~80 launches from one thread).
struct AcrChain {
    AcrProblem pb;
    AcrState* states; AcrHyp* hyps; uint32_t* sorted; double* models;      // two copies each, indexed by launch parity
    uint32_t* best_inliers; uint32_t* index_set;
    unsigned long long* h_word;
    AcrFinish fin;
};
struct AcrChains { AcrChain c[kMaxBatch]; };
// one round of n_chains solves: P3P (one launch) / five-point (two launches); batch_bound and the sort width cover the largest chain
hipError_t launch_acr_round_p3p_chains(const AcrChains& chains, int n_chains, int par, int batch_bound, hipStream_t stream);
hipError_t launch_acr_round_5pt_chains(const AcrChains& chains, int n_chains, int par, int batch_bound, hipStream_t stream);
// batch_bound: iterations the launch grids cover (>= the batch the device state asks for); d_mask .. h_res: where the round that
// completes the run leaves mask / inlier list / result record (device copies + pinned host mirrors)
// the seven-point / four-point models (kind 2: <= 3 per sample, kind 3: 1) of S samples of normalised correspondences: d_out S x M x 9, NaN = no model
hipError_t launch_twoview_minimal(int kind, const double* d_x1, const double* d_x2, int N, const int32_t* d_samples, int S, double* d_out, hipStream_t stream);
// the resection round (and the rounds of kinds 2 / 3, by pb.kind) as ONE launch (replay of the previous round + solve + nfa; acransac.hip): d_states / d_hyps / d_sorted / d_models
// hold two copies, this launch reads copy par ^ 1 and writes copy par; the initial state goes into copy 1 and the first launch has par 0
hipError_t launch_acr_round_p3p(const AcrProblem& pb, int par, AcrState* d_states, AcrHyp* d_hyps, uint32_t* d_sorted, double* d_models,
                                uint32_t* d_best_inliers, uint32_t* d_index_set, unsigned long long* h_word, hipStream_t stream,
                                int batch_bound, uint8_t* d_mask, AcrResult* d_res, uint8_t* h_mask, int32_t* h_inliers, AcrResult* h_res);
// the two-view round as TWO launches (replay of the previous round + samples + five-point solve; nfa), same two-copy scheme
hipError_t launch_acr_round_5pt(const AcrProblem& pb, int par, AcrState* d_states, AcrHyp* d_hyps, uint32_t* d_sorted, double* d_models,
                                uint32_t* d_best_inliers, uint32_t* d_index_set, unsigned long long* h_word, hipStream_t stream,
                                int batch_bound, uint8_t* d_mask, AcrResult* d_res, uint8_t* h_mask, int32_t* h_inliers, AcrResult* h_res);
hipError_t launch_acr_stage(const double* h_pinned, double* d_dst, size_t n_doubles /* even */, hipStream_t stream);
hipError_t launch_acr_stage_chains(const double* const* h_pinned, double* const* d_dst, const size_t* n_doubles /* even */, int n_chains,
                                   hipStream_t stream);

} // namespace clc
#endif
