/*
 * clc_oracle.c -- CPU ORACLE (test infrastructure, see clc_oracle.h for the rules).
 *
 * Scalar restatement of the reference semantics; every function cites what it follows.
 * Build: gcc -O2 -std=c11 -ffp-contract=off (NO -march=native / -mfma), see oracle/Makefile.
 */
#include "clc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* K2NN                                                                                        */
/* ------------------------------------------------------------------------------------------ */

static inline uint64_t ld64(const uint8_t* p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

static inline int hamming512(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int k = 0; k < 8; ++k) d += __builtin_popcountll(ld64(a + 8 * k) ^ ld64(b + 8 * k));
    return d;
}

/* src/CUDAK2NN.cu:54 (initial state), :56-73 (loop), :75 (acceptance; threshold is uint8_t :46) */
void orc_k2nn(const uint8_t* q, int nq, const uint8_t* t, int nt, int threshold,
              int32_t* match_out, uint16_t* best_out, uint16_t* second_out)
{
    const int thr = (int)(uint8_t)threshold;
    for (int i = 0; i < nq; ++i) {
        int best_i = -1, best_v = 100000, second_v = 200000;
        for (int j = 0; j < nt; ++j) {
            const int d = hamming512(q + (size_t)64 * i, t + (size_t)64 * j);
            second_v = d < second_v ? d : second_v;
            if (d < best_v) {
                second_v = best_v;
                best_i = j;
                best_v = d;
            }
        }
        match_out[i] = (nt > 0 && second_v - best_v > thr) ? best_i : -1;
        if (best_out) best_out[i] = (uint16_t)(best_v > 65535 ? 65535 : best_v);
        if (second_out) second_out[i] = (uint16_t)(second_v > 65535 ? 65535 : second_v);
    }
}

/* SURVEY.md 8(a) note N1: order-free definition + exact merge, A = lower train indices. */
void orc_k2nn_split(const uint8_t* q, int nq, const uint8_t* t, int nt, int threshold,
                    int nsplit, int32_t* match_out)
{
    const int thr = (int)(uint8_t)threshold;
    if (nsplit < 1) nsplit = 1;
    const int per = (nt + nsplit - 1) / nsplit;
    for (int i = 0; i < nq; ++i) {
        int A_best = 100000, A_second = 200000, A_idx = -1;
        for (int s = 0; s < nsplit; ++s) {
            const int t0 = s * per, t1 = (t0 + per < nt) ? t0 + per : nt;
            /* partition-local result from the multiset definition */
            int B_best = 100000, B_second = 200000, B_idx = -1;
            for (int j = t0; j < t1; ++j) {
                const int d = hamming512(q + (size_t)64 * i, t + (size_t)64 * j);
                if (d < B_best) { B_second = B_best; B_best = d; B_idx = j; }
                else if (d < B_second) B_second = d;
            }
            if (B_idx < 0) continue;
            if (B_best < A_best) {
                const int sec = A_best < B_second ? A_best : B_second;
                A_best = B_best; A_idx = B_idx; A_second = sec;
            } else {
                A_second = A_second < B_best ? A_second : B_best;
            }
        }
        match_out[i] = (A_idx >= 0 && A_second - A_best > thr) ? A_idx : -1;
    }
}

/* include/coloc/CPUMatcher.hpp:67-76 -> DistanceRatioMatch(0.8f, BRUTE_FORCE_HAMMING, A, B):
 * exhaustive Hamming, top-2 per query, OpenMP over queries.  This is the CPU BASELINE the GPU
 * number is quoted against, so it gets the best popcount the host offers: hardware POPCNT always
 * (the file is built with -mpopcnt), and an AVX-512 VPOPCNTDQ inner loop (one 512-bit descriptor per
 * register) when the CPU has it, selected at run time. */
#if defined(__x86_64__)
#include <immintrin.h>
/* 8 train vectors per iteration: 8 x (xor + vpopcntq) give 8 registers of per-word counts; a
 * 3-stage unpack/permute/add tree transposes and sums them into ONE register holding the 8
 * distances.  The running top-2 is kept per LANE on 64-bit keys (distance << 32 | train index): lane k
 * sees trains j+k; best = min(best, key), second = min(second, max(best, key)); the 8 lanes are merged
 * at the end (keys are unique, so "lowest index wins" comes out of the key order).  No scalar
 * dependency chain in the loop: ~42 vector instructions per 8 comparisons. */
__attribute__((target("avx512f,avx512vpopcntdq"), optimize("O3,unroll-loops")))
static void k2nn_query_avx512(const uint8_t* qrow, const uint8_t* t, int nt, int* best_i, int* best_v, int* second_v)
{
    const __m512i q = _mm512_loadu_si512((const void*)qrow);
    const __m512i ix = _mm512_setr_epi64(0, 1, 8, 9, 4, 5, 12, 13);
    const __m512i iy = _mm512_setr_epi64(2, 3, 10, 11, 6, 7, 14, 15);
    const __m512i none = _mm512_set1_epi64(-1);
    __m512i vbest = none, vsecond = none;
    __m512i vidx = _mm512_setr_epi64(0, 1, 2, 3, 4, 5, 6, 7);
    const __m512i eight = _mm512_set1_epi64(8);
    int j = 0;
    for (; j + 8 <= nt; j += 8) {
        __m512i p[8];
        for (int k = 0; k < 8; ++k)
            p[k] = _mm512_popcnt_epi64(_mm512_xor_si512(q, _mm512_loadu_si512((const void*)(t + (size_t)64 * (j + k)))));
        __m512i s1[4];
        for (int k = 0; k < 4; ++k)      /* [a01 b01 a23 b23 a45 b45 a67 b67] */
            s1[k] = _mm512_add_epi64(_mm512_unpacklo_epi64(p[2 * k], p[2 * k + 1]), _mm512_unpackhi_epi64(p[2 * k], p[2 * k + 1]));
        __m512i s2[2];
        for (int k = 0; k < 2; ++k)      /* [a0123 b0123 c0123 d0123 a4567 b4567 c4567 d4567] */
            s2[k] = _mm512_add_epi64(_mm512_permutex2var_epi64(s1[2 * k], ix, s1[2 * k + 1]),
                                     _mm512_permutex2var_epi64(s1[2 * k], iy, s1[2 * k + 1]));
        const __m512i lo = _mm512_shuffle_i64x2(s2[0], s2[1], 0x44), hi = _mm512_shuffle_i64x2(s2[0], s2[1], 0xEE);
        const __m512i dist = _mm512_add_epi64(lo, hi);                       /* [a b c d e f g h] */
        const __m512i key = _mm512_or_si512(_mm512_slli_epi64(dist, 32), vidx);
        vsecond = _mm512_min_epu64(vsecond, _mm512_max_epu64(vbest, key));
        vbest = _mm512_min_epu64(vbest, key);
        vidx = _mm512_add_epi64(vidx, eight);
    }
    unsigned long long kb[8], ks[8];
    _mm512_storeu_si512((void*)kb, vbest);
    _mm512_storeu_si512((void*)ks, vsecond);
    unsigned long long b = ~0ull, s2nd = ~0ull;
    for (int k = 0; k < 8; ++k) {          /* (b, s) (+) (b', s') = (min, min(max(b, b'), s, s')) */
        const unsigned long long mx = b > kb[k] ? b : kb[k];
        unsigned long long m = s2nd < ks[k] ? s2nd : ks[k];
        m = m < mx ? m : mx;
        b = b < kb[k] ? b : kb[k];
        s2nd = m;
    }
    for (; j < nt; ++j) {                  /* tail */
        const __m512i x = _mm512_xor_si512(q, _mm512_loadu_si512((const void*)(t + (size_t)64 * j)));
        const unsigned long long key = ((unsigned long long)_mm512_reduce_add_epi64(_mm512_popcnt_epi64(x)) << 32) | (unsigned)j;
        const unsigned long long mx = b > key ? b : key;
        s2nd = s2nd < mx ? s2nd : mx;
        b = b < key ? b : key;
    }
    *best_i = b == ~0ull ? -1 : (int)(b & 0xFFFFFFFFu);
    *best_v = b == ~0ull ? 100000 : (int)(b >> 32);
    *second_v = s2nd == ~0ull ? (b == ~0ull ? 200000 : 100000) : (int)(s2nd >> 32);
}
#endif   /* __x86_64__ */

static void k2nn_query_scalar(const uint8_t* qrow, const uint8_t* t, int nt, int* best_i, int* best_v, int* second_v)
{
    uint64_t qq[8];
    for (int k = 0; k < 8; ++k) qq[k] = ld64(qrow + 8 * k);
    int bi = -1, bv = 100000, sv = 200000;
    for (int j = 0; j < nt; ++j) {
        const uint8_t* tp = t + (size_t)64 * j;
        int d = 0;
        for (int k = 0; k < 8; ++k) d += __builtin_popcountll(qq[k] ^ ld64(tp + 8 * k));
        sv = d < sv ? d : sv;
        if (d < bv) { sv = bv; bi = j; bv = d; }
    }
    *best_i = bi; *best_v = bv; *second_v = sv;
}

static double now_s(void)
{
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return (double)clock() / CLOCKS_PER_SEC;
#endif
}

/* 0 = scalar POPCNT, 1 = AVX-512 VPOPCNTDQ.  Chosen once: whichever inner loop is faster on THIS CPU for a
 * 64 x 2048 sample (some hosts execute 512-bit ops at a fraction of the scalar rate). */
static int k2nn_pick_kernel(void)
{
    static int picked = -1;
    if (picked >= 0) return picked;
    int p = 0;
#if defined(__x86_64__)
    if (!getenv("ORC_SCALAR_POPCNT") && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vpopcntdq")) {
        enum { CQ = 64, CT = 2048 };
        uint8_t* buf = (uint8_t*)malloc((size_t)(CQ + CT) * 64);
        uint64_t st = 88172645463325252ull;
        for (size_t i = 0; i < (size_t)(CQ + CT) * 64; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; buf[i] = (uint8_t)(st >> 24); }
        double tm[2] = { 1e30, 1e30 };
        volatile int sink = 0;
        for (int rep = 0; rep < 3; ++rep) {
            for (int w = 0; w < 2; ++w) {
                const double t0 = now_s();
                /* every thread runs the sample at once: what matters is the rate with ALL cores busy */
#ifdef _OPENMP
#pragma omp parallel
#endif
                {
                    int acc = 0;
                    for (int i = 0; i < CQ; ++i) {
                        int bi, bv, sv;
                        if (w) k2nn_query_avx512(buf + (size_t)64 * i, buf + (size_t)64 * CQ, CT, &bi, &bv, &sv);
                        else k2nn_query_scalar(buf + (size_t)64 * i, buf + (size_t)64 * CQ, CT, &bi, &bv, &sv);
                        acc += bi;
                    }
#ifdef _OPENMP
#pragma omp atomic
#endif
                    sink += acc;
                }
                const double dt = now_s() - t0;
                if (dt < tm[w]) tm[w] = dt;
            }
        }
        free(buf);
        p = tm[1] < tm[0] ? 1 : 0;
    }
#endif
    picked = p;
    return p;
}

const char* orc_k2nn_omp_kernel(void) { return k2nn_pick_kernel() ? "avx512-vpopcntdq (auto-selected)" : "scalar popcnt64 (auto-selected)"; }

int orc_k2nn_omp(const uint8_t* q, int nq, const uint8_t* t, int nt, int rule, int threshold,
                 float ratio, int32_t* match_out)
{
    return orc_k2nn_omp_ex(q, nq, t, nt, rule, threshold, ratio, -1, match_out);
}

int orc_k2nn_avx512_available(void)
{
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vpopcntdq");
#else
    return 0;
#endif
}

/* kernel: 0 = the 8 x __builtin_popcountll loop BASELINE.md section 2 specifies ("OpenMVG-equivalent
 * (restated)"), 1 = AVX-512 VPOPCNTDQ (falls back to 0 when the CPU lacks it), -1 = whichever is faster here. */
int orc_k2nn_omp_ex(const uint8_t* q, int nq, const uint8_t* t, int nt, int rule, int threshold,
                    float ratio, int kernel, int32_t* match_out)
{
    const int thr = (int)(uint8_t)threshold;
    const float r2 = ratio * ratio;
    const int wide = kernel < 0 ? k2nn_pick_kernel() : (kernel == 1 && orc_k2nn_avx512_available());
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < nq; ++i) {
        int best_i = -1, best_v = 100000, second_v = 200000;
#if defined(__x86_64__)
        if (wide) k2nn_query_avx512(q + (size_t)64 * i, t, nt, &best_i, &best_v, &second_v);
        else
#endif
            k2nn_query_scalar(q + (size_t)64 * i, t, nt, &best_i, &best_v, &second_v);
        int ok;
        if (rule == 0) ok = (nt > 0) && (second_v - best_v > thr);
        else ok = (nt > 1) && ((float)best_v < r2 * (float)second_v);
        match_out[i] = ok ? best_i : -1;
    }
    return nthreads;
}

/* The same sweep `reps` times inside ONE parallel region, each repetition timed between two team barriers: the figure bench.py's
 * cpu_baseline quotes.  Timing the call from outside puts the team's wake-up into a 4 ms region. */
int orc_k2nn_omp_timed(const uint8_t* q, int nq, const uint8_t* t, int nt, int rule, int threshold, float ratio, int kernel,
                       int reps, int32_t* match_out, double* best_seconds)
{
    const int thr = (int)(uint8_t)threshold;
    const float r2 = ratio * ratio;
    const int wide = kernel < 0 ? k2nn_pick_kernel() : (kernel == 1 && orc_k2nn_avx512_available());
    int nthreads = 1;
    double best = 1e30;
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
#ifdef _OPENMP
#pragma omp single
        nthreads = omp_get_num_threads();
#endif
        for (int r = 0; r < reps; ++r) {
#ifdef _OPENMP
#pragma omp barrier
#endif
            const double t0 = now_s();
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
            for (int i = 0; i < nq; ++i) {
                int best_i = -1, best_v = 100000, second_v = 200000;
#if defined(__x86_64__)
                if (wide) k2nn_query_avx512(q + (size_t)64 * i, t, nt, &best_i, &best_v, &second_v);
                else
#endif
                    k2nn_query_scalar(q + (size_t)64 * i, t, nt, &best_i, &best_v, &second_v);
                int ok;
                if (rule == 0) ok = (nt > 0) && (second_v - best_v > thr);
                else ok = (nt > 1) && ((float)best_v < r2 * (float)second_v);
                match_out[i] = ok ? best_i : -1;
            }                                                     /* (implicit barrier: every thread's share is done) */
            const double t1 = now_s();
#ifdef _OPENMP
#pragma omp master
#endif
            if (t1 - t0 < best) best = t1 - t0;
        }
    }
    *best_seconds = best;
    return nthreads;
}

/* include/coloc/CPUMatcher.hpp:67-76 (computeMatchesPair), :56-65 (matchMapFeatures), :85-89 (matchSceneWithMap):
 * matching::DistanceRatioMatch(ratio, BRUTE_FORCE_HAMMING, regions_I, regions_J, out).  OpenMVG is an empty, unpinned
 * submodule, so this follows SURVEY.md 8(a) row a-9's statement of it (parity unpinned):
 *   regions_I is the DATABASE, regions_J the QUERIES; per query the two smallest Hamming distances over all database
 *   rows; keep iff (float)d1 < ratio*ratio * (float)d2 (ties at the best distance therefore never pass, so the kept
 *   index is unambiguous); emit IndMatch(i_ = database index, j_ = query index); drop exact duplicate pairs; drop
 *   matches whose (x_I, y_I, x_J, y_J) coordinates repeat an earlier one.
 * Fewer than two database rows or no query gives no match.  Output: pairs[2k] = i_, pairs[2k+1] = j_, ordered by
 * (x_I, y_I, x_J, y_J, i_, j_); the order OpenMVG's set-based pass leaves is not restated, so compare as a SET.
 * Returns the match count; *threads_out (optional) = OpenMP threads of the distance sweep. */
typedef struct { float x1, y1, x2, y2; int32_t i, j; } cpm_row;

static int cpm_cmp(const void* a, const void* b)
{
    const cpm_row* p = (const cpm_row*)a; const cpm_row* q = (const cpm_row*)b;
    if (p->x1 != q->x1) return p->x1 < q->x1 ? -1 : 1;
    if (p->y1 != q->y1) return p->y1 < q->y1 ? -1 : 1;
    if (p->x2 != q->x2) return p->x2 < q->x2 ? -1 : 1;
    if (p->y2 != q->y2) return p->y2 < q->y2 ? -1 : 1;
    if (p->i != q->i) return p->i < q->i ? -1 : 1;
    if (p->j != q->j) return p->j < q->j ? -1 : 1;
    return 0;
}

int orc_cpumatcher_pair(const uint8_t* desc_i, const float* xy_i, int n_i, const uint8_t* desc_j, const float* xy_j, int n_j,
                        float ratio, int kernel, int32_t* pairs, int* threads_out)
{
    if (threads_out) *threads_out = 1;
    if (n_i < 2 || n_j <= 0) return 0;
    int32_t* m = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_j);
    cpm_row* rows = (cpm_row*)malloc(sizeof(cpm_row) * (size_t)n_j);
    if (!m || !rows) { free(m); free(rows); return -1; }
    const int nthr = orc_k2nn_omp_ex(desc_j, n_j, desc_i, n_i, 1, 0, ratio, kernel, m);
    if (threads_out) *threads_out = nthr;
    int n = 0;
    for (int j = 0; j < n_j; ++j) {
        if (m[j] < 0) continue;
        const int i = m[j];
        cpm_row r = { xy_i[2 * i], xy_i[2 * i + 1], xy_j[2 * j], xy_j[2 * j + 1], i, j };
        rows[n++] = r;
    }
    qsort(rows, (size_t)n, sizeof(cpm_row), cpm_cmp);
    int k = 0;
    for (int a = 0; a < n; ++a) {
        if (k > 0) {
            const cpm_row* p = &rows[k - 1];
            if (p->i == rows[a].i && p->j == rows[a].j) continue;                                 /* IndMatch::getDeduplicated */
            if (p->x1 == rows[a].x1 && p->y1 == rows[a].y1 && p->x2 == rows[a].x2 && p->y2 == rows[a].y2) continue;   /* IndMatchDecorator */
        }
        rows[k++] = rows[a];
    }
    for (int a = 0; a < k; ++a) { pairs[2 * a] = rows[a].i; pairs[2 * a + 1] = rows[a].j; }
    free(m); free(rows);
    return k;
}

/* ------------------------------------------------------------------------------------------ */
/* pyramid                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* include/coloc/GPUDetector.hpp:109-114 */
void orc_pyramid_dims(uint32_t W, uint32_t H, float scale_factor, int levels,
                      uint32_t* w_out, uint32_t* h_out, float* f_out)
{
    float f = 1.0f;
    w_out[0] = W; h_out[0] = H;
    if (f_out) f_out[0] = 1.0f;
    for (int i = 1; i < levels; ++i) {
        f *= scale_factor;
        w_out[i] = (uint32_t)((float)W / f + 0.5f);
        h_out[i] = (uint32_t)((float)H / f + 0.5f);
        if (f_out) f_out[i] = f;
    }
}

static inline float tap(const uint8_t* img, uint32_t W, uint32_t H, size_t pitch, int x, int y)
{
    /* clamp addressing (GPUDetector.hpp:96-97), normalized-float read (:239): u8 / 255 */
    if (x < 0) x = 0;
    if (y < 0) y = 0;
    if (x > (int)W - 1) x = (int)W - 1;
    if (y > (int)H - 1) y = (int)H - 1;
    return (float)img[(size_t)y * pitch + (size_t)x] / 255.0f;
}

/* src/CUDALERP.cu:157-178.  tex2Dgather(fx+0.5, fy+0.5) returns the bilinear footprint of
 * (fx, fy): w=(i,j) z=(i+1,j) x=(i,j+1) y=(i+1,j+1), i=floor(fx), j=floor(fy). */
void orc_lerp(const uint8_t* img, uint32_t W, uint32_t H, size_t in_pitch, float gxs, float gys,
              uint8_t* out, uint32_t neww, uint32_t newh, size_t out_pitch)
{
    for (uint32_t y = 0; y < newh; ++y) {
        const float fy = ((float)y + 0.5f) * gys - 0.5f;
        const float wt_y = fy - floorf(fy);
        const float invwt_y = 1.0f - wt_y;
        const int j = (int)floorf(fy);
        for (uint32_t x = 0; x < neww; ++x) {
            const float fx = ((float)x + 0.5f) * gxs - 0.5f;
            const int i = (int)floorf(fx);
            const float f_w = tap(img, W, H, in_pitch, i, j);
            const float f_z = tap(img, W, H, in_pitch, i + 1, j);
            const float f_x = tap(img, W, H, in_pitch, i, j + 1);
            const float f_y = tap(img, W, H, in_pitch, i + 1, j + 1);
            const float wt_x = fx - floorf(fx);
            const float invwt_x = 1.0f - wt_x;
            const float xa = invwt_x * f_w + wt_x * f_z;
            const float xb = invwt_x * f_x + wt_x * f_y;
            const float res = 255.0f * (invwt_y * xa + wt_y * xb) + 0.5f;
            out[(size_t)y * out_pitch + x] = (uint8_t)res; /* truncating store, :177 */
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* CLATCH                                                                                      */
/* ------------------------------------------------------------------------------------------ */

static const uint8_t k_latch_pattern[512][6] = {
#include "../coloc_amd/csrc/latch_pattern.inc"
};

const uint8_t* orc_latch_pattern(void) { return &k_latch_pattern[0][0]; }

/* src/CLATCH.cu:161-168: rotated 64x64 window, point sampled with clamp, C truncation of the
 * fp32 coordinate + 0.5f.  Evaluation order exactly as written: (pt.x + (xo*c - yo*s)) + 0.5f. */
void orc_clatch_roi(const uint8_t* level, uint32_t w, uint32_t h, size_t pitch,
                    const orc_keypoint* kp, uint8_t* roi_out)
{
    const float s = (float)sin((double)kp->angle);
    const float c = (float)cos((double)kp->angle);
    for (int r = 0; r < 64; ++r) {
        for (int cc = 0; cc < 64; ++cc) {
            const float xo = (float)(cc - 32);
            const float yo = (float)(r - 32);
            const float fx = ((float)kp->x + (xo * c - yo * s)) + 0.5f;
            const float fy = ((float)kp->y + (xo * s + yo * c)) + 0.5f;
            int sx = (int)fx, sy = (int)fy;
            if (sx < 0) sx = 0;
            if (sy < 0) sy = 0;
            if (sx > (int)w - 1) sx = (int)w - 1;
            if (sy > (int)h - 1) sy = (int)h - 1;
            roi_out[r * 64 + cc] = level[(size_t)sy * pitch + (size_t)sx];
        }
    }
}

/* src/CLATCH.cu:169-188: S_n = sum over the 8x8 patch of (A-B)^2 - (C-B)^2, bit = S_n < 0,
 * bit n&31 of little-endian uint32 word n>>5. */
void orc_clatch(const uint8_t* const* levels, const uint32_t* w, const uint32_t* h,
                const size_t* pitch, const orc_keypoint* kps, int n, uint8_t* desc_out)
{
    uint8_t roi[64 * 64];
    for (int k = 0; k < n; ++k) {
        const int lv = kps[k].scale;
        orc_clatch_roi(levels[lv], w[lv], h[lv], pitch[lv], &kps[k], roi);
        uint32_t words[16];
        memset(words, 0, sizeof words);
        for (int t = 0; t < 512; ++t) {
            const uint8_t* p = k_latch_pattern[t];
            int32_t S = 0;
            for (int dy = 0; dy < 8; ++dy) {
                for (int dx = 0; dx < 8; ++dx) {
                    const int32_t A = roi[(p[0] + dy) * 64 + p[1] + dx];
                    const int32_t B = roi[(p[2] + dy) * 64 + p[3] + dx];
                    const int32_t C = roi[(p[4] + dy) * 64 + p[5] + dx];
                    S += (A - B) * (A - B) - (C - B) * (C - B);
                }
            }
            if (S < 0) words[t >> 5] |= 1u << (t & 31);
        }
        for (int wd = 0; wd < 16; ++wd) {
            uint8_t* o = desc_out + (size_t)64 * k + 4 * wd;
            o[0] = (uint8_t)(words[wd]);
            o[1] = (uint8_t)(words[wd] >> 8);
            o[2] = (uint8_t)(words[wd] >> 16);
            o[3] = (uint8_t)(words[wd] >> 24);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* host feeders                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* include/coloc/FeatureAngle.h:160-177 */
static float orc_fast_atan2(float y, float x)
{
    const float PI = 3.1415927f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (ax >= ay) {
        const float c = ay / (ax + FLT_MIN);
        const float cc = c * c;
        a = (((-0.0443265555479f * cc + 0.1555786518f) * cc - 0.325808397f) * cc + 0.9997878412f) * c;
    } else {
        const float c = ax / (ay + FLT_MIN);
        const float cc = c * c;
        a = PI * 0.5f - (((-0.0443265555479f * cc + 0.1555786518f) * cc - 0.325808397f) * cc + 0.9997878412f) * c;
    }
    if (x < 0.0f) a = PI - a;
    if (y < 0.0f) a = -a;
    return a;
}

/* include/coloc/FeatureAngle.h:179-246: 37-pixel disc (rows of 3,5,7,7,7,5,3), x weight = column
 * offset, y weight = row offset; int16 sums (|sum| <= 26*255, no wrap). */
float orc_feature_angle(const uint8_t* img, int px, int py, int step)
{
    static const int half[7] = { 1, 2, 3, 3, 3, 2, 1 };
    int xs = 0, ys = 0;
    for (int r = -3; r <= 3; ++r) {
        const int hw = half[r + 3];
        for (int c = -hw; c <= hw; ++c) {
            const int v = img[(size_t)(py + r) * step + (px + c)];
            xs += c * v;
            ys += r * v;
        }
    }
    return orc_fast_atan2((float)(int16_t)ys, (float)(int16_t)xs);
}

/* include/coloc/KFAST.h:164-500.  The reference walks each row in 32-column blocks with a
 * 16-column "retreat" when the low half of the cardinal pre-test mask is empty (:259-265) and
 * masks the final partial block with (1 << (cols-j-3)) - 1 (:245).  When the walk lands exactly
 * on j == cols-35 that shift count is 32, which x86 evaluates as a shift by 0 -> mask 0 -> the
 * last 32 columns of that row are dropped.  The walk is reproduced here (scalar per pixel inside
 * a block) so that the restatement matches what the compiled reference produces. */
static const int k_ring_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int k_ring_dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

static int fast_pretest(const uint8_t* p, int stride, int t)
{
    const int c = *p;
    const int hi = c + t > 255 ? 255 : c + t;
    const int lo = c - t < 0 ? 0 : c - t;
    const int p9 = p[3 * stride], p5 = p[3], p1 = p[-3 * stride], p13 = p[-3];
    const int b = ((p9 > hi) & (p5 > hi)) | ((p5 > hi) & (p1 > hi)) | ((p1 > hi) & (p13 > hi)) | ((p13 > hi) & (p9 > hi));
    const int d = ((p9 < lo) & (p5 < lo)) | ((p5 < lo) & (p1 < lo)) | ((p1 < lo) & (p13 < lo)) | ((p13 < lo) & (p9 < lo));
    return b | d;
}

static int fast_is_corner(const uint8_t* p, int stride, int t, uint8_t* score)
{
    const int c = *p;
    const int hi = c + t > 255 ? 255 : c + t;
    const int lo = c - t < 0 ? 0 : c - t;
    int ring[24];
    for (int k = 0; k < 24; ++k) ring[k] = p[k_ring_dy[k & 15] * stride + k_ring_dx[k & 15]];
    int bc = 0, dc = 0, bmax = 0, dmax = 0;
    for (int k = 0; k < 24; ++k) {
        bc = ring[k] > hi ? bc + 1 : 0;
        dc = ring[k] < lo ? dc + 1 : 0;
        if (bc > bmax) bmax = bc;
        if (dc > dmax) dmax = dc;
    }
    if ((bmax > dmax ? bmax : dmax) <= 8) return 0;
    /* corner score :300-374: max over the 16 arcs of 9 of max(min(p-ring), -max(p-ring)) */
    int best = -32768;
    for (int s = 0; s < 16; ++s) {
        int mn = 32767, mx = -32768;
        for (int k = s; k < s + 9; ++k) {
            const int v = c - ring[k];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        const int val = mn > -mx ? mn : -mx;
        if (val > best) best = val;
    }
    *score = (uint8_t)best;
    return 1;
}

int orc_fast9(const uint8_t* img, int cols, int rows, int stride, uint8_t threshold,
              orc_keypoint* out, int cap)
{
    int count = 0;
    if (cols < 7 || rows < 7) return 0;
    uint8_t* buf = (uint8_t*)calloc((size_t)3 * cols, 1);
    uint8_t* rowbuf[3] = { buf, buf + cols, buf + 2 * cols };
    for (int i = 3; i < rows - 2; ++i) {
        uint8_t* cur = rowbuf[i % 3];
        memset(cur, 0, (size_t)cols);
        if (i < rows - 3) {
            const uint8_t* row = img + (size_t)i * stride;
            int j = 3;
            for (;;) {
                const int full = j < cols - 35;
                int width = 32;
                if (!full) {
                    const int sh = (cols - j - 3) & 31; /* x86 shift-count masking of :245 */
                    width = (cols - j - 3 == 32) ? 0 : sh;
                    (void)sh;
                }
                uint32_t m = 0;
                for (int x = 0; x < (full ? 32 : width); ++x)
                    if (fast_pretest(row + j + x, stride, threshold)) m |= 1u << x;
                if (m != 0) {
                    if (full && (m & 0xFFFFu) == 0) {
                        j -= 16; /* :259-265 retreat; net +16 after the loop increment */
                    } else {
                        for (int x = 0; x < (full ? 32 : width); ++x) {
                            uint8_t sc;
                            if (fast_is_corner(row + j + x, stride, threshold, &sc)) cur[j + x] = sc;
                        }
                    }
                }
                if (!full) break;
                j += 32;
            }
        }
        if (i == 3) continue;
        const uint8_t* last = rowbuf[(i - 1) % 3];
        const uint8_t* last2 = rowbuf[(i - 2) % 3];
        for (int j = 3; j < cols - 3; ++j) {
            const uint8_t sc = last[j];
            if (!sc) continue;
            if ((sc > last[j - 1]) & (sc > last[j + 1]) & (sc > cur[j - 1]) & (sc > cur[j]) & (sc > cur[j + 1]) &
                (sc > last2[j - 1]) & (sc > last2[j]) & (sc > last2[j + 1])) {
                if (count < cap) {
                    out[count].x = j;
                    out[count].y = i - 1;
                    out[count].score = sc;
                    out[count].angle = 0.0f;
                    out[count].scale = 0;
                }
                ++count;
            }
        }
    }
    free(buf);
    return count < cap ? count : cap;
}

/* include/coloc/GPUDetector.hpp:172-179 */
void orc_features_from_kps(const orc_keypoint* kps, int n, float* feat_out)
{
    for (int i = 0; i < n; ++i) {
        const float scale = (float)pow((double)1.2f, (double)kps[i].scale); /* std::pow(float,uint8) -> double */
        feat_out[4 * i + 0] = scale * (float)kps[i].x;
        feat_out[4 * i + 1] = scale * (float)kps[i].y;
        feat_out[4 * i + 2] = 7.0f * scale;
        feat_out[4 * i + 3] = kps[i].angle;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* PnP residuals                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* include/coloc/Localizer.hpp:61-72 builds pt3D (3xN) / pt2D (2xN); the residual OpenMVG's
 * resection kernel evaluates is the squared pixel reprojection error of P = K[R|t]. */
void orc_pnp_residuals(const double* Rt, int H, const double* X, const double* x, int N,
                       const double* K, double* err_out)
{
    for (int h = 0; h < H; ++h) {
        const double* P = Rt + (size_t)12 * h;
        for (int i = 0; i < N; ++i) {
            const double Xw = X[3 * i + 0], Yw = X[3 * i + 1], Zw = X[3 * i + 2];
            const double xc = ((P[0] * Xw + P[1] * Yw) + P[2] * Zw) + P[3];
            const double yc = ((P[4] * Xw + P[5] * Yw) + P[6] * Zw) + P[7];
            const double zc = ((P[8] * Xw + P[9] * Yw) + P[10] * Zw) + P[11];
            const double u = (K[0] * xc + K[1] * yc) + K[2] * zc;
            const double v = (K[3] * xc + K[4] * yc) + K[5] * zc;
            const double w = (K[6] * xc + K[7] * yc) + K[8] * zc;
            const double du = x[2 * i + 0] - u / w;
            const double dv = x[2 * i + 1] - v / w;
            err_out[(size_t)h * N + i] = du * du + dv * dv;
        }
    }
}

void orc_pnp_score(const double* err, int H, int N, double thr2, int32_t* count_out,
                   double* cost_out)
{
    for (int h = 0; h < H; ++h) {
        int32_t cnt = 0;
        double cost = 0.0;
        for (int i = 0; i < N; ++i) {
            const double e = err[(size_t)h * N + i];
            if (e < thr2) { ++cnt; cost += e; }
            else cost += thr2;
        }
        if (count_out) count_out[h] = cnt;
        if (cost_out) cost_out[h] = cost;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* two-view scoring                                                                            */
/* ------------------------------------------------------------------------------------------ */

void orc_epipolar_residuals(const double* F, int H, const double* x1, const double* x2, int N, double* err_out)
{
    for (int h = 0; h < H; ++h) {
        const double* f = F + (size_t)9 * h;
        for (int i = 0; i < N; ++i) {
            const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
            const double a0 = (f[0] * u1 + f[1] * v1) + f[2];      /* F x1 */
            const double a1 = (f[3] * u1 + f[4] * v1) + f[5];
            const double a2 = (f[6] * u1 + f[7] * v1) + f[8];
            const double b0 = (f[0] * u2 + f[3] * v2) + f[6];      /* F^T x2 */
            const double b1 = (f[1] * u2 + f[4] * v2) + f[7];
            const double d = (u2 * a0 + v2 * a1) + a2;             /* x2^T F x1 */
            err_out[(size_t)h * N + i] = (d * d) * (1.0 / (a0 * a0 + a1 * a1) + 1.0 / (b0 * b0 + b1 * b1)) / 4.0;
        }
    }
}
