// clc_acr.h -- arithmetic shared by the host and device halves of the a-contrario RANSAC (AC-RANSAC) path:
// a portable log10, the per-iteration sampler and the NFA term.  Plain C/C++ that compiles for the host and, through
// CLC_ACR_HD, for gfx950; the same source gives the same bits on both (IEEE fp64 +,-,*,/ only, no FMA contraction --
// the library and the oracle are built with -ffp-contract=off).
//
// What it serves: openMVG::robust::ACRANSAC as called by SfM_Localizer::Localize (reference
// include/coloc/Localizer.hpp:82-93: error_max = +inf, max_iteration = 256) and by RobustMatcher::filterEssential
// (include/coloc/RobustMatcher.hpp:161-171).  OpenMVG is an empty submodule in the reference tree, so the algorithm is
// restated from its publication: L. Moisan, P. Moulon, P. Monasse, "Automatic Homographic Registration of a Pair of
// Images, with A Contrario Elimination of Outliers", IPOL 2012 (NFA(k) = (n - m) * C(n, k) * C(k, m) * alpha_k^(k - m),
// minimised over the k smallest residuals of every model).
#ifndef CLC_ACR_H
#define CLC_ACR_H

#include <stdint.h>
#include <string.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __HIPCC__
#define CLC_ACR_HD __host__ __device__ inline
#else
#define CLC_ACR_HD static inline
#endif

// ---- log10 --------------------------------------------------------------------------------------------------------
// The NFA of a model is a sum of log10 terms and two models are compared with a strict '<' (bestNFA / ACRANSAC), so the
// host and the device must produce the SAME double for log10(e): libm on the host and the device math library differ
// in the last bits.  This is the classic table-free algorithm (argument reduction to [sqrt(2)/2, sqrt(2)], s = f / (2 + f),
// degree-14 even polynomial in s; log10 = n log10(2) + ln(x) / ln(10) with split constants), within 2 ulp of libm (99 % equal);
// tests/test_acransac.py compares it with libm over 10^6 arguments.
CLC_ACR_HD double clc_acr_from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
CLC_ACR_HD uint64_t clc_acr_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

// natural logarithm of a positive, finite, NORMAL x in [0.5, 2)-ish range after reduction; general positive x accepted
CLC_ACR_HD double clc_acr_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = clc_acr_bits(x);
    int32_t hx = (int32_t)(u >> 32);
    int32_t k = 0;
    if (hx < 0x00100000) {                       // subnormal (zero / negative are excluded by the callers): scale up
        k -= 54;
        x *= 1.80143985094819840000e+16;
        u = clc_acr_bits(x);
        hx = (int32_t)(u >> 32);
    }
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i = (hx + 0x95f64) & 0x100000;
    u = (u & 0xFFFFFFFFull) | ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32);     // x or x / 2 in [sqrt(2)/2, sqrt(2))
    x = clc_acr_from_bits(u);
    k += i >> 20;
    const double f = x - 1.0;
    const double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {           // |f| < 2^-20
        if (f == 0.0) return k == 0 ? 0.0 : dk * ln2_hi + dk * ln2_lo;
        const double R = f * f * (0.5 - 0.33333333333333333 * f);
        return k == 0 ? f - R : dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    if (((hx - 0x6147a) | (0x6b851 - hx)) > 0) {
        const double hfsq = 0.5 * f * f;
        return k == 0 ? f - (hfsq - s * (hfsq + R)) : dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    return k == 0 ? f - s * (f - R) : dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// log10 of a positive finite x
CLC_ACR_HD double clc_acr_log10(double x)
{
    const double ivln10 = 4.34294481903251816668e-01, log10_2hi = 3.01029995663611771306e-01,
                 log10_2lo = 3.69423907715893078616e-13;
    uint64_t u = clc_acr_bits(x);
    int32_t hx = (int32_t)(u >> 32);
    int32_t k = 0;
    if (hx < 0x00100000) {
        k -= 54;
        x *= 1.80143985094819840000e+16;
        u = clc_acr_bits(x);
        hx = (int32_t)(u >> 32);
    }
    k += (hx >> 20) - 1023;
    const int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
    const double y = (double)(k + i);
    u = (u & 0xFFFFFFFFull) | ((uint64_t)(uint32_t)hx << 32);
    const double z = y * log10_2lo + ivln10 * clc_acr_log(clc_acr_from_bits(u));
    return z + y * log10_2hi;
}

// ---- sampler ------------------------------------------------------------------------------------------------------
// OpenMVG draws the minimal sample of iteration `iter` with UniformSample over the current index set from a
// std::mt19937 stream -- a sequential stream whose bits are not pinned by anything the reference holds.  Here the
// sample is a pure function of (seed, iteration, size of the index set): a counter-based splitmix64 stream per
// iteration, positions = high half of (32 random bits x n_index) (no 64-bit division on the GPU), drawn with rejection
// of repeats.  That makes the result independent of how many iterations are evaluated speculatively in one batch on
// the GPU, and lets the sequential oracle reproduce it.
CLC_ACR_HD uint64_t clc_acr_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// m (<= 8) distinct positions in [0, n_index), n_index > m
CLC_ACR_HD void clc_acr_sample(uint64_t seed, uint32_t iter, uint32_t n_index, int m, uint32_t* pos)
{
    uint64_t s = clc_acr_mix(seed ^ clc_acr_mix((uint64_t)iter + 1u));
    for (int j = 0; j < m; ++j) {
        uint32_t p;
        bool again;
        do {
            s = clc_acr_mix(s);
            p = (uint32_t)(((s >> 32) * (uint64_t)n_index) >> 32);
            again = false;
            for (int q = 0; q < j; ++q) again = again || pos[q] == p;
        } while (again);
        pos[j] = p;
    }
}

// the same draw with the sample size as a template argument: every index is a compile-time constant, so on the GPU the positions
// stay in registers (with a run-time m the array is dynamically indexed private memory: 48 B of scratch in the select kernel,
// a memory round trip per access in a kernel that is nothing but a latency chain)
#if defined(__cplusplus)
template <int M>
CLC_ACR_HD void clc_acr_sample_t(uint64_t seed, uint32_t iter, uint32_t n_index, uint32_t (&pos)[M])
{
    uint64_t s = clc_acr_mix(seed ^ clc_acr_mix((uint64_t)iter + 1u));
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < M; ++j) {
        uint32_t p;
        bool again;
        do {
            s = clc_acr_mix(s);
            p = (uint32_t)(((s >> 32) * (uint64_t)n_index) >> 32);
            again = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int q = 0; q < M; ++q) again = again || (q < j && pos[q] == p);
        } while (again);
        pos[j] = p;
    }
}
#endif

// ---- NFA term -----------------------------------------------------------------------------------------------------
// log10 NFA of taking the k smallest residuals as inliers (bestNFA in OpenMVG's robust_estimator_ACRansac.hpp):
//   loge0 + (logalpha0 + mult * log10(e_k + FLT_EPSILON)) * (k - m) + logc_n[k] + logc_k[k],  evaluated left to right,
// e_k = the k-th smallest residual, logc_n[k] = log10 C(n, k), logc_k[k] = log10 C(k, m) as floats.
CLC_ACR_HD double clc_acr_nfa(double loge0, double logalpha0, double mult, double e_k, int k, int m, float logc_n_k, float logc_k_k)
{
    const double logalpha = logalpha0 + mult * clc_acr_log10(e_k + 1.1920928955078125e-07);
    return ((loge0 + logalpha * (double)(k - m)) + (double)logc_n_k) + (double)logc_k_k;
}

#endif
