// clc_sincos.h -- deterministic sin/cos of an fp32 angle, identical on host (g++) and device (hipcc).
//
// The CLATCH window rotation (reference src/CLATCH.cu:161 `sin(pt.angle), cos(pt.angle)`) is the
// one place where the descriptor depends on a transcendental.  CUDA's sinf/cosf bits cannot be
// reproduced, so the contract is: s = fp32(sin(angle)), c = fp32(cos(angle)),
// i.e. the correctly rounded single-precision value.  This header gets there by evaluating in
// fp64 (Cody-Waite reduction by pi/2 + the classic degree-13/14 minimax kernels, |err| < 1 ulp of
// double) and rounding once to fp32; the result can differ from the exactly rounded value only
// when the true value lies within ~2^-29 ulp(fp32) of a rounding boundary.  The oracle
// computes the same quantity independently with libm in double; tests/test_sincos.py compares
// the two over tens of millions of angles.
//
// The two polynomials are evaluated with EXPLICIT fused multiply-adds (`__builtin_fma`: one correctly
// rounded operation, identical on every IEEE machine -- v_fma_f64 on the device, hardware or libm fma on
// the host), which halves the fp64 instruction count of the per-keypoint evaluation; everything else is
// plain +, -, * and conversions, and implicit contraction stays off (-ffp-contract=off + the pragma below).
#ifndef CLC_SINCOS_H
#define CLC_SINCOS_H

#if defined(__HIPCC__) || defined(__HIP__)
#define CLC_HD __host__ __device__ inline
#else
#define CLC_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

CLC_HD void clc_sincosf(float angle, float* s_out, float* c_out)
{
    const double x = (double)angle;
    // k = nearest integer to x * 2/pi (ties irrelevant: any consistent choice is exact below)
    const double two_over_pi = 6.36619772367581382433e-01;
    const double kd0 = x * two_over_pi;
    const double kd1 = kd0 + (kd0 < 0.0 ? -0.5 : 0.5);
    // clamp so the int conversion is defined for any finite fp32 (|x| <= 3.4e38 is nonsense for an
    // angle; accuracy is only claimed for |x| <= 1e5, determinism for everything)
    const double kd2 = kd1 > 2.0e9 ? 2.0e9 : (kd1 < -2.0e9 ? -2.0e9 : kd1);
    const int k = (int)kd2;
    const double kd = (double)k;
    // pi/2 split: hi has 33 significant bits so kd*hi is exact for |k| < 2^20
    const double pio2_1 = 1.57079632673412561417e+00;
    const double pio2_1t = 6.07710050650619224932e-11;
    const double r = (x - kd * pio2_1) - kd * pio2_1t;
    const double z = r * r;
    // sin(r) on [-pi/4, pi/4]
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = S6;
    ps = __builtin_fma(ps, z, S5);
    ps = __builtin_fma(ps, z, S4);
    ps = __builtin_fma(ps, z, S3);
    ps = __builtin_fma(ps, z, S2);
    ps = __builtin_fma(ps, z, S1);
    const double sr = __builtin_fma(r * z, ps, r);
    // cos(r) on [-pi/4, pi/4]
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = C6;
    pc = __builtin_fma(pc, z, C5);
    pc = __builtin_fma(pc, z, C4);
    pc = __builtin_fma(pc, z, C3);
    pc = __builtin_fma(pc, z, C2);
    pc = __builtin_fma(pc, z, C1);
    const double cr = __builtin_fma(z * z, pc, 1.0 - 0.5 * z);
    double sv, cv;
    switch (k & 3) {
        case 0: sv = sr; cv = cr; break;
        case 1: sv = cr; cv = -sr; break;
        case 2: sv = -sr; cv = -cr; break;
        default: sv = -cr; cv = sr; break;
    }
    *s_out = (float)sv;
    *c_out = (float)cv;
}

#endif
