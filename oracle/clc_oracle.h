/*
 * clc_oracle.h -- CPU ORACLE for the CoLoC describe -> match -> pose-scoring hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, link, import or execute anything under oracle/.
 * The product (coloc_amd/, libcoloc_hip.so) never includes or links this directory.
 *
 * It is a plain-C, scalar, single-thread RESTATEMENT of the source-level semantics of the
 * reference's three CUDA kernels and of the host conventions around them.  Every function
 * cites the reference file:line it follows.  Floating point is IEEE fp32/fp64 evaluated in
 * the order written in the reference source, with NO fused multiply-add contraction (build
 * with -ffp-contract=off and without -mfma; see oracle/Makefile).
 *
 * PARITY UNPINNED (kernels): the reference ships no tests, golden vectors or fixtures for this
 * path and its CUDA kernels cannot be compiled here (no nvcc / CUDA
 * runtime / NVIDIA device), so this restatement cannot be checked against reference OUTPUT.
 * It is checked instead by (i) an independent numpy brute force (tests/test_oracle_k2nn.py),
 * (ii) lane-level emulations of the reference's 32-lane butterflies written from the source
 * (tests/lane_emulation.py) and (iii) the committed golden fixtures under tests/golden/, which
 * freeze the restatement's own outputs.  The only reference code that DOES build here is the
 * dependency-free host feeder pair KFAST.h + FeatureAngle.h; oracle/Makefile compiles it from
 * /root/reference into oracle/_ref/ and tests pin orc_fast9 / orc_feature_angle against it.
 */
#ifndef CLC_ORACLE_H
#define CLC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Keypoint.h:155-163 -- 20 bytes, level-local integer pixel coordinates. */
typedef struct orc_keypoint {
    int32_t x;
    int32_t y;
    uint8_t score;
    float   angle;
    uint8_t scale;
} orc_keypoint;

/* ---- K2NN (src/CUDAK2NN.cu:46-75) ---------------------------------------------------- */

/* Sequential restatement of the per-query loop: train vectors visited in index order,
 * best_v=100000 / second_v=200000 initial state, strict '<' keeps the earliest index,
 * result = (second_v - best_v > (uint8_t)threshold) ? best_i : -1.
 * q, t: rows of 64 bytes (8 little-endian uint64).  best_out/second_out may be NULL; they
 * receive min(value, 65535) so the 100000/200000 sentinels saturate.
 * nt == 0 is undefined in the reference (uninitialised best_i, CUDAK2NN.cu:54,75); the
 * oracle defines it as "no match" (-1). */
void orc_k2nn(const uint8_t* q, int nq, const uint8_t* t, int nt, int threshold,
              int32_t* match_out, uint16_t* best_out, uint16_t* second_out);

/* Same result computed from the order-free definition of SURVEY.md 8(a) note N1 with the train
 * set cut into `nsplit` contiguous parts and the exact two-partition merge applied left to
 * right -- the decomposition the HIP kernel uses.  Used to test the merge rule on the CPU. */
void orc_k2nn_split(const uint8_t* q, int nq, const uint8_t* t, int nt, int threshold,
                    int nsplit, int32_t* match_out);

/* CPU baseline matcher, OpenMP over queries (include/coloc/CPUMatcher.hpp:67-76 calls
 * openMVG::matching::DistanceRatioMatch(0.8f, BRUTE_FORCE_HAMMING, ...); OpenMVG is an empty,
 * unpinned submodule, so this is a restatement of brute-force Hamming top-2 + acceptance).
 * rule 0: K2NN rule (second - best > threshold), rule 1: ratio rule best < ratio^2 * second.
 * Returns the number of threads used. */
int orc_k2nn_omp(const uint8_t* q, int nq, const uint8_t* t, int nt, int rule, int threshold,
                 float ratio, int32_t* match_out);
/* which inner loop the auto-selection uses on this CPU */
const char* orc_k2nn_omp_kernel(void);
/* explicit inner loop: 0 = 8 x __builtin_popcountll per pair ("OpenMVG-equivalent
 * (restated)"), 1 = AVX-512 VPOPCNTDQ best effort (falls back to 0 if absent), -1 = auto */
int orc_k2nn_omp_ex(const uint8_t* q, int nq, const uint8_t* t, int nt, int rule, int threshold,
                    float ratio, int kernel, int32_t* match_out);
int orc_k2nn_avx512_available(void);

/* The whole CPU comparator of include/coloc/CPUMatcher.hpp:56-98: DistanceRatioMatch(ratio, BRUTE_FORCE_HAMMING,
 * regions_I = database, regions_J = queries) = top-2 sweep + ratio test + IndMatch(i_ = database, j_ = query) + the two
 * de-duplication passes (identical pairs; identical (x,y,x,y)).  xy_* are n x 2 float positions.  pairs holds 2*n_j
 * int32; returns the match count (compare as a set -- see the definition). */
int orc_cpumatcher_pair(const uint8_t* desc_i, const float* xy_i, int n_i, const uint8_t* desc_j, const float* xy_j, int n_j,
                        float ratio, int kernel, int32_t* pairs, int* threads_out);

/* ---- pyramid (include/coloc/GPUDetector.hpp:109-114,249-254; src/CUDALERP.cu:157-178) --- */

/* Level dims: f_0 = 1, f_i = f_{i-1} * scale_factor (fp32); w_i = (uint32)((float)W / f_i + 0.5f). */
void orc_pyramid_dims(uint32_t W, uint32_t H, float scale_factor, int levels,
                      uint32_t* w_out, uint32_t* h_out, float* f_out);

/* Bilinear resample of the level-0 image into one level (always from level 0, gxs = gys = f). */
void orc_lerp(const uint8_t* img, uint32_t W, uint32_t H, size_t in_pitch, float gxs, float gys,
              uint8_t* out, uint32_t neww, uint32_t newh, size_t out_pitch);

/* ---- CLATCH (src/CLATCH.cu:157-188, include/coloc/CLATCH.h:170) -------------------------- */

/* levels[i] points at level i (u8, pitch[i] bytes per row, w[i] x h[i]); clamp addressing.
 * desc_out: n x 64 bytes (16 little-endian uint32; bit n&31 of word n>>5 = (S_n < 0)).
 * sin/cos: s = (float)sin((double)angle), c = (float)cos((double)angle) -- the correctly
 * rounded fp32 value, computed here with libm in double and rounded once (the reference's
 * CUDA sinf/cosf bits are unknowable). */
void orc_clatch(const uint8_t* const* levels, const uint32_t* w, const uint32_t* h,
                const size_t* pitch, const orc_keypoint* kps, int n, uint8_t* desc_out);

/* The 64 x 64 sampled window of one keypoint (row stride 64), for ROI-level tests. */
void orc_clatch_roi(const uint8_t* level, uint32_t w, uint32_t h, size_t pitch,
                    const orc_keypoint* kp, uint8_t* roi_out);

/* triplet table re-encoded as 512 x {a_row,a_col,b_row,b_col,c_row,c_col}; returns pointer to
 * 3072 bytes (for the sha256 check against CLATCH.h:170). */
const uint8_t* orc_latch_pattern(void);

/* ---- host feeders (include/coloc/KFAST.h:164-540, FeatureAngle.h:160-246) --------------- */

/* FAST-9 with corner score and strict 3x3 non-max suppression, valid region cols [3,cols-3),
 * rows [3,rows-3) minus the NMS border; keypoints emitted in (row, col) ascending order, which
 * is the order the reference's band-concatenation yields.  Returns the count (<= cap). */
int orc_fast9(const uint8_t* img, int cols, int rows, int stride, uint8_t threshold,
              orc_keypoint* out, int cap);

/* Intensity-centroid orientation over the 37-pixel disc with the 7th-order fastAtan2. */
float orc_feature_angle(const uint8_t* img, int px, int py, int step);

/* ---- conversion to regions (include/coloc/GPUDetector.hpp:167-182) ---------------------- */

/* feat_out: n x {x, y, scale, orientation} floats = {s*x, s*y, 7*s, angle}, s = pow(1.2f, scale). */
void orc_features_from_kps(const orc_keypoint* kps, int n, float* feat_out);

/* ---- PnP residuals (include/coloc/Localizer.hpp:59-108 plumbing) -------- */

/* Rt: H x 12 doubles, row-major [R|t] (3 rows of 4).  X: N x 3 (Eigen 3xN column-major),
 * x: N x 2 pixel observations.  K: 9 doubles row-major.  err_out[h*N+i] = squared pixel
 * reprojection error || x_i - hnormalized(K (R_h X_i + t_h)) ||^2 in fp64. */
void orc_pnp_residuals(const double* Rt, int H, const double* X, const double* x, int N,
                       const double* K, double* err_out);

/* Per-hypothesis inlier count and truncated cost sum_i min(err, thr2) (MSAC score). */
void orc_pnp_score(const double* err, int H, int N, double thr2, int32_t* count_out,
                   double* cost_out);

/* ---- two-view scoring (include/coloc/RobustMatcher.hpp:153-186) ---------- */

/* Symmetric epipolar distance of H fundamental matrices (row-major 9) over N pixel correspondences,
 * the error model RobustMatcher::filterEssential hands to AC-RANSAC
 * (openMVG::fundamental::kernel::SymmetricEpipolarDistanceError; OpenMVG is absent, formula restated
 * from its published definition): e = (x2^T F x1)^2 (1/|(F x1)_xy|^2 + 1/|(F^T x2)_xy|^2) / 4. */
void orc_epipolar_residuals(const double* F, int H, const double* x1, const double* x2, int N, double* err_out);

/* ---- a-contrario RANSAC (oracle/clc_oracle_acr.c; include/coloc/Localizer.hpp:82-93, RobustMatcher.hpp:153-171) ---- */

/* minimal solver callback: `sample` holds the m data indices (3 for kind 0, 5 for kind 1, 7 for kind 2, 4 for kind 3); writes up to
 * 4 (10, 3, 1) models of 12 (18, 9, 9) doubles -- [R|t] row-major, {F (9), E (9)}, resp. F / H IN NORMALISED COORDINATES
 * (orc_tv_normalize) -- in solver order and returns how many */
typedef int (*orc_acr_fit_fn)(void* user, const uint32_t* sample, double* models_out);

/* kind 0: resection, a = X (N x 3), b = x (N x 2 pixels), K1 = intrinsics (9, row-major);
 * kind 1: essential, a = x1, b = x2 (N x 2 pixels), img_w x img_h = size of image 2 (logalpha0);
 * kind 2: fundamental (RobustMatcher.hpp:128-151), kind 3: homography (:188-239): a = x1, b = x2 (pixels), both images img_w x img_h;
 *         the loop runs on coordinates conditioned by the image size and model_out is the matrix brought back to pixels.
 * precision = +inf: pure a-contrario mode (what the reference passes).  Outputs: the model, the inliers in ascending
 * residual order (capacity N), the precision found (pixels for kind 0), the minimum log10 NFA, the iteration that
 * produced the model and the number of iterations run.  Returns 1 if a meaningful model (NFA < 0) was found. */
int orc_acransac(int kind, const double* a, const double* b, int n, const double* K1, int img_w, int img_h,
                 int max_iteration, uint64_t seed, double precision, orc_acr_fit_fn fit, void* user,
                 double* model_out, uint32_t* inliers_out, int* n_inliers_out, double* error_max_out, double* min_nfa_out,
                 int32_t* best_iter_out, int32_t* iterations_run_out);
/* log10 C(n, k) and log10 C(k, m) float tables, k = 0..n */
void orc_acr_tables(int n, int m, float* logc_n, float* logc_k);
/* min_k NFA(k) of one model given its residuals (kernel units); *k_out = the minimising k */
double orc_acr_best_nfa(const double* err, int n, int m, int max_models, double logalpha0, double mult, int* k_out);
/* the sample of iteration `iter`: m distinct positions in [0, n_index) -- the oracle's OWN statement of the documented sampler */
void orc_acr_sample(uint64_t seed, uint32_t iter, uint32_t n_index, int m, uint32_t* pos);


/* ---- the seven-point / four-point models (oracle/clc_oracle_twoview.c; RobustMatcher.hpp:128-151, :188-239) ---- */
/* q1, q2: 7 (4) normalised correspondences {u, v}; F_out: up to 3 matrices of 9 (x2^T F x1 = 0), H_out: 9 (x2 ~ H x1); returns the count */
int orc_seven_point(const double* q1, const double* q2, double* F_out);
int orc_four_point(const double* q1, const double* q2, double* H_out);
/* ACKernelAdaptor's conditioning by the image size: {d, tx, ty} of T = [d 0 tx; 0 d ty; 0 0 1]; x_n = T x; F = T^T Fn T, H = T^-1 Hn T */
void orc_tv_normalizer(int w, int h, double* d_tx_ty);
void orc_tv_normalize(int w, int h, const double* x, int n, double* xn);
void orc_tv_unnormalize(int homography, int w, int h, const double* Mn, double* M);
/* kind 2: squared distance of x2 to the epipolar line F x1; kind 3: squared transfer error |x2 - H x1|^2 (normalised coordinates) */
void orc_tv_residuals(int kind, const double* M, const double* q1, const double* q2, int n, double* e);

#ifdef __cplusplus
}
#endif
#endif
