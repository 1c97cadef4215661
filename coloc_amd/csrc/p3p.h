// p3p.h -- minimal perspective-three-point solver, one problem per lane (host + device inline).
//
// Role: hypothesis generator for the batched PnP RANSAC (clc_pnp_ransac).  The reference asks
// OpenMVG for resection::SolverType::P3P_KE_CVPR17 inside AC-RANSAC (include/coloc/Localizer.hpp:93);
// OpenMVG is an empty, unpinned submodule in the reference snapshot, so no line of that solver is
// available to follow.  Any exact minimal solver yields the same pose set; this one is the classic
// distance formulation (Grunert 1841 / Haralick et al. 1994) derived from scratch:
//   unknown depths s1, s2 = u s1, s3 = v s1 along the unit bearings f1, f2, f3;
//   law of cosines on the three point pairs; eliminating s1 and u leaves a quartic in v whose
//   coefficients are BUILT BY POLYNOMIAL ARITHMETIC (no closed-form coefficient table to mistype):
//       q = (a^2 - c^2) / b^2,   N(v) = (q-1) v^2 - 2 q cos(beta) v + q + 1,   D(v) = 2 (cos(gamma) - v cos(alpha)),
//       W(v) = 1 + v^2 - 2 v cos(beta),     u = N / D,
//       b^2 (D^2 + N^2 - 2 cos(gamma) N D) - c^2 D^2 W = 0.
//   The quartic is solved in closed form (Ferrari, trigonometric / Cardano resolvent) and each real
//   root is polished with Newton steps on the original polynomial; the rigid transform follows from
//   the two orthonormal triads spanned by the three points in the world and in the camera frame.
// fp64 throughout.
#ifndef CLC_P3P_H
#define CLC_P3P_H

#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define P3P_HD __host__ __device__ inline
#else
#define P3P_HD static inline
#endif

// The solver is one long dependent chain per lane (a P3P launch is 128 x 4 lanes: latency, not throughput), so it is written
// for few instructions: fused multiply-adds and reciprocals by v_rcp_f64 + two Newton steps (1 ulp) instead of the IEEE division
// sequence (57 divisions per solve, ~10 instructions each).
// Round 5: every fusion is SPELLED (fma()) and the file is compiled with contraction OFF.  Rounds 2-4 built it under
// `#pragma clang fp contract(fast)`, which lets the compiler fuse a multiply into an add wherever it likes -- per inlining context:
// p3p_sample_root is force-inlined into p3p_kernel (whose poses the tests hand to the a-contrario oracle) and into acr_round_kernel
// (which solves its own samples), and the "winning pose == the oracle's, bit for bit" tests held only while both contexts happened
// to get the same choices.  With contraction off an expression is exactly the IEEE operations written here, in
// this order, wherever it is inlined: the two kernels agree by construction, not by coincidence.
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double p3p_rcp(const double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
#else
P3P_HD double p3p_rcp(const double x) { return 1.0 / x; }
#endif

// real roots of x^3 + a x^2 + b x + c: returns count (1 or 3)
P3P_HD int p3p_solve_cubic(double a, double b, double c, double* x)
{
    const double a3 = a * (1.0 / 3.0);
    const double p = fma(-a, a3, b);
    const double q = fma(2.0 * a3 * a3, a3, fma(-a3, b, c));
    const double p3 = p * p * p * (1.0 / 27.0);
    const double disc = fma(0.25 * q, q, p3);
    if (disc > 0.0) {
        const double s = sqrt(disc);
        const double u = cbrt(-0.5 * q + s), v = cbrt(-0.5 * q - s);
        x[0] = u + v - a3;
        return 1;
    }
    const double r = sqrt(-p3);
    double cosphi = r > 0.0 ? -0.5 * q * p3p_rcp(r) : 0.0;
    cosphi = cosphi > 1.0 ? 1.0 : (cosphi < -1.0 ? -1.0 : cosphi);
    const double phi = acos(cosphi);
    const double m = 2.0 * sqrt(-p * (1.0 / 3.0));
    x[0] = fma(m, cos(phi * (1.0 / 3.0)), -a3);
    x[1] = fma(m, cos((phi + 2.0 * M_PI) * (1.0 / 3.0)), -a3);
    x[2] = fma(m, cos((phi + 4.0 * M_PI) * (1.0 / 3.0)), -a3);
    return 3;
}

// real roots of c4 x^4 + c3 x^3 + c2 x^2 + c1 x + c0 (c4 != 0) in closed form, UNPOLISHED (p3p_polish_root).
// Fixed slots, no compaction (a running count would index the output dynamically = scratch memory on the device):
// roots[0], roots[1] come from the first quadratic factor, roots[2], roots[3] from the second; returns the mask of
// the slots that hold a real root.
P3P_HD int p3p_solve_quartic(const double* co, double* roots)
{
    const double i4 = p3p_rcp(co[4]);
    const double a = co[3] * i4, b = co[2] * i4, c = co[1] * i4, d = co[0] * i4;
    // depressed quartic y^4 + p y^2 + q y + r, x = y - a/4
    const double a2 = a * a;
    const double p = fma(-0.375, a2, b);
    const double q = fma(0.125 * a2, a, fma(-0.5 * a, b, c));
    const double r = fma(-(3.0 / 256.0) * a2, a2, fma(0.0625 * a2, b, fma(-0.25 * a, c, d)));
    int mask = 0;
    double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
    if (fabs(q) < 1e-14 * (1.0 + fabs(p) + fabs(r))) {
        // biquadratic
        const double disc = fma(p, p, -4.0 * r);
        if (disc >= 0.0) {
            const double s = sqrt(disc);
            const double z0 = 0.5 * (-p + s), z1 = 0.5 * (-p - s);
            if (z0 >= 0.0) { y0 = sqrt(z0); y1 = -y0; mask |= 3; }
            if (z1 >= 0.0) { y2 = sqrt(z1); y3 = -y2; mask |= 12; }
        }
    } else {
        // resolvent: m^3 + p m^2 + (p^2/4 - r) m - q^2/8 = 0, take the largest real root (it is > 0)
        double m3[3];
        const int nm = p3p_solve_cubic(p, fma(0.25 * p, p, -r), -0.125 * q * q, m3);
        double m = m3[0];
        if (nm == 3) { m = m3[1] > m ? m3[1] : m; m = m3[2] > m ? m3[2] : m; }
        if (m > 0.0) {
            const double s = sqrt(2.0 * m);
            const double t0 = fma(0.5, p, m);
            const double t1 = q * p3p_rcp(2.0 * s);
            // y^2 + s y + (t0 - t1) = 0  and  y^2 - s y + (t0 + t1) = 0
            double disc = fma(s, s, -4.0 * (t0 - t1));
            if (disc >= 0.0) { const double sq = sqrt(disc); y0 = 0.5 * (-s + sq); y1 = 0.5 * (-s - sq); mask |= 3; }
            disc = fma(s, s, -4.0 * (t0 + t1));
            if (disc >= 0.0) { const double sq = sqrt(disc); y2 = 0.5 * (s + sq); y3 = 0.5 * (s - sq); mask |= 12; }
        }
    }
    const double a4 = 0.25 * a;
    roots[0] = y0 - a4; roots[1] = y1 - a4; roots[2] = y2 - a4; roots[3] = y3 - a4;
    return mask;
}

// Newton polish of one root on the original polynomial
P3P_HD double p3p_polish_root(const double* co, double x)
{
    for (int it = 0; it < 3; ++it) {
        const double f = fma(fma(fma(fma(co[4], x, co[3]), x, co[2]), x, co[1]), x, co[0]);
        const double df = fma(fma(fma(4.0 * co[4], x, 3.0 * co[3]), x, 2.0 * co[2]), x, co[1]);
        if (df == 0.0) break;
        x = fma(-f, p3p_rcp(df), x);
    }
    return x;
}

P3P_HD void p3p_cross(const double* a, const double* b, double* c)
{
    c[0] = fma(a[1], b[2], -(a[2] * b[1]));
    c[1] = fma(a[2], b[0], -(a[0] * b[2]));
    c[2] = fma(a[0], b[1], -(a[1] * b[0]));
}
P3P_HD double p3p_dot(const double* a, const double* b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
P3P_HD bool p3p_normalize(double* a)
{
    const double n = sqrt(p3p_dot(a, a));
    if (!(n > 1e-300)) return false;
    const double in = p3p_rcp(n);
    a[0] *= in; a[1] *= in; a[2] *= in;
    return true;
}
// orthonormal triad (columns e1,e2,e3) of three points
P3P_HD bool p3p_triad(const double* A, const double* B, const double* C, double E[3][3])
{
    double e1[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] };
    double w[3] = { C[0] - A[0], C[1] - A[1], C[2] - A[2] };
    double e3[3], e2[3];
    if (!p3p_normalize(e1)) return false;
    p3p_cross(e1, w, e3);
    if (!p3p_normalize(e3)) return false;
    p3p_cross(e3, e1, e2);
    for (int i = 0; i < 3; ++i) { E[i][0] = e1[i]; E[i][1] = e2[i]; E[i][2] = e3[i]; }
    return true;
}

// Everything of a P3P problem that does not depend on WHICH root is being turned into a pose.
struct P3PProblem {
    double N[3], D[2], W[3], co[5];   // u = N(v) / D(v), W(v), the quartic in v
    double b2;
    double E[3][3];                   // world triad
    double roots[4];                  // closed-form roots of the quartic (unpolished) / of the cubic fallback
    int mask;                         // which of the four slots hold a real root
    bool polish;                      // quartic branch: Newton-polish a root before use
};

// X: three world points (3x3 row-major, one point per row); f: three UNIT bearing vectors.  Returns false if degenerate.
P3P_HD bool p3p_prepare(const double X[3][3], const double f[3][3], P3PProblem& p)
{
    double d12[3], d13[3], d23[3];
    for (int i = 0; i < 3; ++i) { d12[i] = X[0][i] - X[1][i]; d13[i] = X[0][i] - X[2][i]; d23[i] = X[1][i] - X[2][i]; }
    const double c2 = p3p_dot(d12, d12), b2 = p3p_dot(d13, d13), a2 = p3p_dot(d23, d23);
    p.mask = 0;
    if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return false;
    const double ca = p3p_dot(f[1], f[2]), cb = p3p_dot(f[0], f[2]), cg = p3p_dot(f[0], f[1]);
    const double q = (a2 - c2) * p3p_rcp(b2);
    // polynomials in v, ascending coefficients
    p.N[0] = q + 1.0; p.N[1] = -2.0 * q * cb; p.N[2] = q - 1.0;
    p.D[0] = 2.0 * cg; p.D[1] = -2.0 * ca;
    p.W[0] = 1.0; p.W[1] = -2.0 * cb; p.W[2] = 1.0;
    p.b2 = b2;
    const double* N = p.N; const double* D = p.D; const double* W = p.W;
    double DD[3] = { D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1] };
    double NN[5] = { 0, 0, 0, 0, 0 }, ND[4] = { 0, 0, 0, 0 }, DDW[5] = { 0, 0, 0, 0, 0 };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) NN[i + j] = fma(N[i], N[j], NN[i + j]);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) ND[i + j] = fma(N[i], D[j], ND[i + j]);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) DDW[i + j] = fma(DD[i], W[j], DDW[i + j]);
    for (int k = 0; k < 5; ++k) {
        const double dd = k < 3 ? DD[k] : 0.0, nd = k < 4 ? ND[k] : 0.0;
        p.co[k] = fma(b2, fma(-2.0 * cg, nd, dd + NN[k]), -(c2 * DDW[k]));
    }
    const double* co = p.co;
    const double scale = fabs(co[0]) + fabs(co[1]) + fabs(co[2]) + fabs(co[3]) + fabs(co[4]);
    if (!(scale > 0.0)) return false;
    if (fabs(co[4]) > 1e-12 * scale) {
        p.mask = p3p_solve_quartic(co, p.roots);
        p.polish = true;
    } else if (fabs(co[3]) > 1e-12 * scale) {
        p.roots[3] = 0.0;
        const double i3 = p3p_rcp(co[3]);
        p.mask = (1 << p3p_solve_cubic(co[2] * i3, co[1] * i3, co[0] * i3, p.roots)) - 1;
        p.polish = false;
    } else {
        return false;
    }
    return p3p_triad(X[0], X[1], X[2], p.E);
}

// Root k of a prepared problem -> pose P (12 doubles, row-major [R|t], x_cam = R X + t).  Returns false if the root
// does not give a valid pose.
P3P_HD bool p3p_pose_from_root(const P3PProblem& p, const double X[3][3], const double f[3][3], const int k, double* P)
{
    if (!((p.mask >> k) & 1)) return false;
    const double rk = k == 0 ? p.roots[0] : (k == 1 ? p.roots[1] : (k == 2 ? p.roots[2] : p.roots[3]));   // no dynamic index
    const double v = p.polish ? p3p_polish_root(p.co, rk) : rk;
    if (!(v > 0.0)) return false;
    const double den = fma(p.D[1], v, p.D[0]);
    if (fabs(den) < 1e-12) return false;
    const double u = fma(fma(p.N[2], v, p.N[1]), v, p.N[0]) * p3p_rcp(den);
    if (!(u > 0.0)) return false;
    const double w = fma(fma(p.W[2], v, p.W[1]), v, p.W[0]);
    if (!(w > 0.0)) return false;
    const double s1 = sqrt(p.b2 * p3p_rcp(w)), s2 = u * s1, s3 = v * s1;
    const double Q0[3] = { s1 * f[0][0], s1 * f[0][1], s1 * f[0][2] };
    const double Q1[3] = { s2 * f[1][0], s2 * f[1][1], s2 * f[1][2] };
    const double Q2[3] = { s3 * f[2][0], s3 * f[2][1], s3 * f[2][2] };
    double G[3][3];
    if (!p3p_triad(Q0, Q1, Q2, G)) return false;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) P[4 * i + j] = fma(G[i][2], p.E[j][2], fma(G[i][1], p.E[j][1], G[i][0] * p.E[j][0]));   // R = G E^T
    for (int i = 0; i < 3; ++i) P[4 * i + 3] = fma(-P[4 * i + 2], X[0][2], fma(-P[4 * i + 1], X[0][1], fma(-P[4 * i], X[0][0], Q0[i])));
    bool finite = true;
    for (int i = 0; i < 12; ++i) finite = finite && (P[i] == P[i]) && fabs(P[i]) < 1e300;
    return finite;
}

// All poses of one problem, compacted: Rt_out receives up to 4 poses of 12 doubles.  Returns the count.
P3P_HD int p3p_solve(const double X[3][3], const double f[3][3], double* Rt_out)
{
    P3PProblem p;
    if (!p3p_prepare(X, f, p)) return 0;
    int ns = 0;
    for (int k = 0; k < 4; ++k)
        if (p3p_pose_from_root(p, X, f, k, Rt_out + 12 * ns)) ++ns;
    return ns;
}

#if defined(__HIPCC__) || defined(__HIP__)
// One root of one sample, straight from the correspondence arrays: pose slot `root` (12 doubles [R|t], NaNs when the root has no
// valid pose or a sample index is out of range) of the P3P problem on points i0, i1, i2 of (X: N x 3 world points, x: N x 2 pixels,
// K: 3 x 3 row-major).  Shared by p3p_kernel (the hypothesis generator the tests hand to the a-contrario oracle) and
// acr_round_kernel (which solves its own sample in place); `out` may point to global memory or to LDS.  Both inline it (a call to one
// not-inlined body costs 140-200 B of callee-saved spills per lane and 6 us per solve); since round 5 every fused multiply-add in here
// is written out and nothing else may be contracted (top of the file), so the two agree bit for bit by construction -- and every
// a-contrario test still compares the winning pose bit for bit with the oracle's, which gets its poses from p3p_kernel.
static __device__ __forceinline__ void p3p_sample_root(const double* X, const double* x, const double* K, const int i0, const int i1,
                                                       const int i2, const int N, const int root, double* out)
{
    double Xs[3][3], f[3][3];
    bool ok = true;
    const double fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    const double ifx = p3p_rcp(fx), ify = p3p_rcp(fy);
    const int ids[3] = { i0, i1, i2 };
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int p = 0; p < 3; ++p) {
        int i = ids[p];
        if (i < 0 || i >= N) { ok = false; i = 0; }
        Xs[p][0] = X[3 * i]; Xs[p][1] = X[3 * i + 1]; Xs[p][2] = X[3 * i + 2];
        const double yn = (x[2 * i + 1] - cy) * ify;
        const double xn = fma(-sk, yn, x[2 * i] - cx) * ifx;
        const double inrm = p3p_rcp(sqrt(fma(xn, xn, fma(yn, yn, 1.0))));
        f[p][0] = xn * inrm; f[p][1] = yn * inrm; f[p][2] = inrm;
    }
    P3PProblem prob;
    ok = ok && p3p_prepare(Xs, f, prob);
    double P[12];
    const bool have = ok && p3p_pose_from_root(prob, Xs, f, root, P);
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e < 12; ++e) out[e] = have ? P[e] : qnan;
}
#endif

#endif
