/*
 * clc_oracle_acr.c -- CPU ORACLE (test infrastructure, see clc_oracle.h) for the a-contrario RANSAC the reference runs
 * through OpenMVG:
 *   - pose:      SfM_Localizer::Localize(P3P_KE_CVPR17, ..., {error_max = +inf, max_iteration = 256})
 *                reference include/coloc/Localizer.hpp:82-93  ->  robust::ACRANSAC(ACKernelAdaptorResection_Intrinsics)
 *   - two-view:  robust::ACRANSAC(ACKernelAdaptorEssential<FivePointSolver, SymmetricEpipolarDistanceError>, inliers,
 *                256, &E, +inf)   reference include/coloc/RobustMatcher.hpp:153-171
 *   - and the same loop over ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError> ('F', :128-151) and
 *                ACKernelAdaptor<FourPointSolver, AsymmetricError> ('H', :188-239): kinds 2 / 3, see clc_oracle_twoview.c
 *
 * PARITY UNPINNED: lib/openMVG is an empty, unpinned submodule in the reference tree (.gitmodules:1-3), nothing of it
 * can be built or run here and the reference holds no fixture for this step.  The control flow and the NFA arithmetic
 * below are therefore a restatement of the PUBLISHED algorithm -- L. Moisan, P. Moulon, P. Monasse, "Automatic
 * Homographic Registration of a Pair of Images, with A Contrario Elimination of Outliers", IPOL 2012, and the structure
 * of openMVG/robust_estimation/robust_estimator_ACRansac.hpp as publicly documented (OpenMVG >= 1.1: float log-combination
 * tables built from a float log10 table; 10 % of the iterations reserved for sampling among the inliers of the first
 * meaningful model):
 *
 *   n data, m = minimal sample size, M = max models per sample
 *   loge0 = log10(M (n - m));  logc_n[k] = log10 C(n, k), logc_k[k] = log10 C(k, m)  (float tables)
 *   reserve = max_iter / 10;  n_iter = max_iter - reserve;  index set = all data
 *   for iter < n_iter:
 *       sample m distinct elements of the index set -> models (<= M)
 *       for each model in solver order: residuals over ALL data, sorted ascending with their indices;
 *           NFA(k) = loge0 + (logalpha0 + mult log10(e_(k) + FLT_EPSILON)) (k - m) + logc_n[k] + logc_k[k],  k = m+1 .. n
 *           (while e_(k) <= max threshold); the model's value is min_k NFA(k) (first k on ties);
 *           if it is < min_nfa: min_nfa, model, inliers = the k first indices of the sorted order, error_max = e_(k)
 *       if (improved and min_nfa < 0) or (last iteration and reserve left):
 *           no inliers at all yet: one more iteration, reserve - 1
 *           else: index set = inliers; if reserve left: n_iter = iter + 1 + reserve, reserve = 0
 *   min_nfa >= 0 -> no model.
 *
 * What differs from OpenMVG by necessity: (1) the random sample of an iteration is NOT std::mt19937 + UniformSample
 * (bits unknowable here) but a documented pure function of (seed, iteration, index-set size), STATED HERE ON ITS OWN
 * (orc_acr_sample below; the product's statement is coloc_amd/csrc/clc_acr.h -- tests/test_acransac.py holds the two
 * against each other); (2) the minimal solver is a callback -- the tests plug in hypotheses from the GPU kernels, or
 * from HOST builds of the solvers (tests/host/p3p_host_lib.cpp, fivept_host_lib.cpp), so that this oracle checks exactly
 * what it restates: residual normalisation, ordering, NFA, model selection, the phase switch and the inlier set.
 * log10 is libm's, everywhere (round 6: this file shares NO code with the product -- it used to include clc_acr.h for
 * the sampler and for a portable log10; the product's own log10 differs from libm's by at most 2 ulp, so the discrete
 * results -- model, inlier list, iteration count -- must be identical and the NFA agrees to ~1e-15 relative).
 */
#include "clc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- the sample of an iteration: specification, restated -----------------------------------------------------------------
 * stream:  z_0 = mix(seed XOR mix(iter + 1)),  z_{t+1} = mix(z_t),  with splitmix64's output function
 *          mix(z): z += 0x9E3779B97F4A7C15;  z = (z XOR z >> 30) * 0xBF58476D1CE4E5B9;  z = (z XOR z >> 27) * 0x94D049BB133111EB;
 *                  return z XOR z >> 31
 * draw t:  position = floor( (z_t >> 32) * n_index / 2^32 )   (the high half of a 32 x 32 -> 64-bit product)
 * sample:  the first m DISTINCT positions in draw order (a draw that repeats an earlier position is skipped). */
static uint64_t orc_splitmix(uint64_t z)
{
    z = z + UINT64_C(0x9E3779B97F4A7C15);
    z ^= z >> 30; z *= UINT64_C(0xBF58476D1CE4E5B9);
    z ^= z >> 27; z *= UINT64_C(0x94D049BB133111EB);
    z ^= z >> 31;
    return z;
}
void orc_acr_sample(uint64_t seed, uint32_t iter, uint32_t n_index, int m, uint32_t* pos)
{
    uint64_t z = orc_splitmix(seed ^ orc_splitmix((uint64_t)iter + 1));
    int have = 0;
    while (have < m) {
        z = orc_splitmix(z);
        const uint32_t r = (uint32_t)(z >> 32);
        const uint32_t p = (uint32_t)(((uint64_t)r * (uint64_t)n_index) / (UINT64_C(1) << 32));
        int seen = 0;
        for (int q = 0; q < have; ++q) seen |= pos[q] == p;
        if (!seen) pos[have++] = p;
    }
}

typedef struct { double e; uint32_t i; } err_index;

static int cmp_err_index(const void* a, const void* b)
{
    const err_index* x = (const err_index*)a;
    const err_index* y = (const err_index*)b;
    if (x->e < y->e) return -1;
    if (x->e > y->e) return 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);      /* std::sort on pair<double, uint32_t>: lexicographic */
}

/* log10 C(n, k) with a float log10 table and float accumulation (OpenMVG's logcombi<float>) */
static float logcombi(uint32_t k, uint32_t n, const float* lg)
{
    if (k >= n || k == 0) return 0.0f;
    if (n - k < k) k = n - k;
    float r = 0.0f;
    for (uint32_t i = 1; i <= k; ++i) r += lg[n - i + 1] - lg[i];
    return r;
}

void orc_acr_tables(int n, int m, float* logc_n, float* logc_k)
{
    float* lg = (float*)malloc(sizeof(float) * (size_t)(n + 2));
    lg[0] = 0.0f;
    for (int k = 1; k <= n + 1; ++k) lg[k] = (float)log10((double)k);
    /* logcombi(k, n) re-sums the same prefix for every k: O(n^2) as written in OpenMVG, a running prefix here -- the
     * additions and their order are identical, so are the floats */
    float* prefix = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    prefix[0] = 0.0f;
    for (int i = 1; i <= n; ++i) prefix[i] = prefix[i - 1] + (lg[n - i + 1] - lg[i]);
    for (int k = 0; k <= n; ++k) {
        uint32_t kk = (uint32_t)k;
        if (kk >= (uint32_t)n || kk == 0) { logc_n[k] = 0.0f; continue; }
        if ((uint32_t)n - kk < kk) kk = (uint32_t)n - kk;
        logc_n[k] = prefix[kk];
    }
    for (int k = 0; k <= n; ++k) logc_k[k] = logcombi((uint32_t)m, (uint32_t)k, lg);
    free(prefix);
    free(lg);
}

/* residuals of one model over all data, in the units the kernel adaptor hands to ACRANSAC */
static void acr_errors(int kind, const double* model, const double* a, const double* b, int n, const double* K1, double* e)
{
    if (kind == 0) {
        /* ACKernelAdaptorResection_Intrinsics::Errors: (camera residual in pixels * N1(0,0)).squaredNorm(),
         * N1(0,0) = imagePlane_toCameraPlaneError(1) = 1 / focal */
        const double s = 1.0 / K1[0];
        for (int i = 0; i < n; ++i) {
            const double Xw = a[3 * i + 0], Yw = a[3 * i + 1], Zw = a[3 * i + 2];
            const double xc = ((model[0] * Xw + model[1] * Yw) + model[2] * Zw) + model[3];
            const double yc = ((model[4] * Xw + model[5] * Yw) + model[6] * Zw) + model[7];
            const double zc = ((model[8] * Xw + model[9] * Yw) + model[10] * Zw) + model[11];
            const double u = (K1[0] * xc + K1[1] * yc) + K1[2] * zc;
            const double v = (K1[3] * xc + K1[4] * yc) + K1[5] * zc;
            const double w = (K1[6] * xc + K1[7] * yc) + K1[8] * zc;
            const double du = (b[2 * i + 0] - u / w) * s;
            const double dv = (b[2 * i + 1] - v / w) * s;
            e[i] = du * du + dv * dv;
        }
    } else if (kind == 1) {
        orc_epipolar_residuals(model, 1, a, b, n, e);     /* SymmetricEpipolarDistanceError on pixels, F = model[0..8] */
    } else {
        orc_tv_residuals(kind, model, a, b, n, e);        /* 'F' / 'H' on the normalised coordinates (clc_oracle_twoview.c) */
    }
}

int orc_acransac(int kind, const double* a, const double* b, int n, const double* K1, int img_w, int img_h,
                 int max_iteration, uint64_t seed, double precision, orc_acr_fit_fn fit, void* user,
                 double* model_out, uint32_t* inliers_out, int* n_inliers_out, double* error_max_out, double* min_nfa_out,
                 int32_t* best_iter_out, int32_t* iterations_run_out)
{
    /* kind 0 resection (P3P), 1 essential (five points), 2 fundamental (seven points), 3 homography (four points) */
    static const int k_m[4] = { 3, 5, 7, 4 }, k_M[4] = { 4, 10, 3, 1 }, k_md[4] = { 12, 18, 9, 9 };
    if (kind < 0 || kind > 3) return 0;
    const int m = k_m[kind], M = k_M[kind], md = k_md[kind];
    if (n_inliers_out) *n_inliers_out = 0;
    if (error_max_out) *error_max_out = 0.0;
    if (min_nfa_out) *min_nfa_out = INFINITY;
    if (best_iter_out) *best_iter_out = -1;
    if (iterations_run_out) *iterations_run_out = 0;
    if (n <= m) return 0;
    /* resection: log10(pi) (error on the normalised camera plane); essential: point-to-line, 2 D / A * 0.5 of image 2 */
    double logalpha0, mult, norm2;
    double* norm_pts = NULL;
    double nrm0 = 1.0;                                   /* N2(0,0) of the ACKernelAdaptor kinds */
    if (kind == 0) { logalpha0 = log10(M_PI); mult = 1.0; norm2 = (1.0 / K1[0]) * (1.0 / K1[0]); }
    else if (kind >= 2) {
        /* ACKernelAdaptor: both point sets conditioned by the image size; point-to-line ('F') log10(2 D / A / N2(0,0)) with the
         * square root of the error, point-to-point ('H') log10(pi / A / N2(0,0)^2); the models live in normalised coordinates */
        double t[3];
        orc_tv_normalizer(img_w, img_h, t);
        nrm0 = t[0];
        const double D = sqrt((double)img_w * (double)img_w + (double)img_h * (double)img_h), A = (double)img_w * (double)img_h;
        if (kind == 2) { logalpha0 = log10(2.0 * D / A / nrm0); mult = 0.5; }
        else { logalpha0 = log10(M_PI / A / (nrm0 * nrm0)); mult = 1.0; }
        norm2 = nrm0 * nrm0;
        norm_pts = (double*)malloc(sizeof(double) * 4 * (size_t)n);
        orc_tv_normalize(img_w, img_h, a, n, norm_pts);
        orc_tv_normalize(img_w, img_h, b, n, norm_pts + 2 * (size_t)n);
        a = norm_pts;
        b = norm_pts + 2 * (size_t)n;
    } else {
        const double D = sqrt((double)img_w * (double)img_w + (double)img_h * (double)img_h), A = (double)img_w * (double)img_h;
        const double al = 2.0 * D / A * .5;
        logalpha0 = log10(al);
        mult = 0.5;
        norm2 = 1.0;
    }
    const double max_threshold = isinf(precision) ? INFINITY : precision * norm2;
    const double loge0 = log10((double)M * (double)(n - m));
    float* logc_n = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    float* logc_k = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    orc_acr_tables(n, m, logc_n, logc_k);
    double* e = (double*)malloc(sizeof(double) * (size_t)n);
    err_index* se = (err_index*)malloc(sizeof(err_index) * (size_t)n);
    uint32_t* index = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    uint32_t* inl = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n);
    double* models = (double*)malloc(sizeof(double) * (size_t)(M * md));
    for (int i = 0; i < n; ++i) index[i] = (uint32_t)i;
    uint32_t n_index = (uint32_t)n;
    int n_inl = 0;
    double min_nfa = INFINITY, error_max = INFINITY;
    int reserve = max_iteration / 10;
    int n_iter = max_iteration - reserve;
    int ac_mode = isinf(precision);
    int iter = 0;
    for (; iter < n_iter; ++iter) {
        uint32_t pos[8], sample[8];
        orc_acr_sample(seed, (uint32_t)iter, n_index, m, pos);
        for (int j = 0; j < m; ++j) sample[j] = index[pos[j]];
        const int nm = fit(user, sample, models);
        int better = 0;
        for (int k = 0; k < nm; ++k) {
            const double* model = models + (size_t)k * md;
            acr_errors(kind, model, a, b, n, K1, e);
            if (!ac_mode) {
                int cnt = 0;
                for (int i = 0; i < n; ++i) cnt += e[i] <= max_threshold;
                if (cnt > 2.5 * m) ac_mode = 1;
            }
            if (!ac_mode) continue;
            for (int i = 0; i < n; ++i) { se[i].e = e[i]; se[i].i = (uint32_t)i; }
            qsort(se, (size_t)n, sizeof(err_index), cmp_err_index);
            double best_nfa = INFINITY;
            int best_k = m;
            for (int kk = m + 1; kk <= n && se[kk - 1].e <= max_threshold; ++kk) {
                const double logalpha = logalpha0 + mult * log10(se[kk - 1].e + (double)FLT_EPSILON);
                const double nfa = ((loge0 + logalpha * (double)(kk - m)) + (double)logc_n[kk]) + (double)logc_k[kk];
                if (nfa < best_nfa) { best_nfa = nfa; best_k = kk; }
            }
            if (best_nfa < min_nfa) {
                better = 1;
                min_nfa = best_nfa;
                n_inl = best_k;
                for (int i = 0; i < best_k; ++i) inl[i] = se[i].i;
                error_max = se[best_k - 1].e;
                memcpy(model_out, model, sizeof(double) * (size_t)md);
                if (best_iter_out) *best_iter_out = iter;
            }
        }
        if ((better && min_nfa < 0) || (iter + 1 == n_iter && reserve)) {
            if (n_inl == 0) { n_iter++; reserve--; }
            else {
                memcpy(index, inl, sizeof(uint32_t) * (size_t)n_inl);
                n_index = (uint32_t)n_inl;
                if (reserve) { n_iter = iter + 1 + reserve; reserve = 0; }
            }
        }
    }
    if (iterations_run_out) *iterations_run_out = iter;
    if (min_nfa >= 0) n_inl = 0;
    if (n_inl > 0) {
        memcpy(inliers_out, inl, sizeof(uint32_t) * (size_t)n_inl);
        /* unormalizeError: resection sqrt(e) / N1(0,0) (pixels); essential: identity */
        if (error_max_out) *error_max_out = kind == 0 ? sqrt(error_max) / (1.0 / K1[0]) : (kind == 1 ? error_max : sqrt(error_max) / nrm0);
        if (kind >= 2) {                                 /* Unnormalize(model): back to pixels */
            double Mp[9];
            orc_tv_unnormalize(kind == 3, img_w, img_h, model_out, Mp);
            memcpy(model_out, Mp, sizeof Mp);
        }
    }
    if (n_inliers_out) *n_inliers_out = n_inl;
    if (min_nfa_out) *min_nfa_out = min_nfa;
    free(models); free(inl); free(index); free(se); free(e); free(logc_k); free(logc_n); free(norm_pts);
    return n_inl > 0;
}

/* the NFA scan of ONE model on caller-supplied residuals (already in kernel units): for unit tests */
double orc_acr_best_nfa(const double* err, int n, int m, int max_models, double logalpha0, double mult, int* k_out)
{
    float* logc_n = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    float* logc_k = (float*)malloc(sizeof(float) * (size_t)(n + 1));
    orc_acr_tables(n, m, logc_n, logc_k);
    err_index* se = (err_index*)malloc(sizeof(err_index) * (size_t)n);
    for (int i = 0; i < n; ++i) { se[i].e = err[i]; se[i].i = (uint32_t)i; }
    qsort(se, (size_t)n, sizeof(err_index), cmp_err_index);
    const double loge0 = log10((double)max_models * (double)(n - m));
    double best = INFINITY;
    int bk = m;
    for (int kk = m + 1; kk <= n; ++kk) {
        const double nfa = ((loge0 + (logalpha0 + mult * log10(se[kk - 1].e + (double)FLT_EPSILON)) * (double)(kk - m)) + (double)logc_n[kk]) + (double)logc_k[kk];
        if (nfa < best) { best = nfa; bk = kk; }
    }
    if (k_out) *k_out = bk;
    free(se); free(logc_k); free(logc_n);
    return best;
}
