// twoview_min.h -- the two other minimal solvers RobustMatcher can put under its a-contrario filter (reference
// include/coloc/RobustMatcher.hpp:128-151 and :188-239, selected by params->model at :399-405):
//     'F'  openMVG::fundamental::kernel::SevenPointSolver   7 correspondences -> <= 3 fundamental matrices
//     'H'  openMVG::homography::kernel::FourPointSolver     4 correspondences -> 1 homography
// both on the coordinates ACKernelAdaptor has normalised by the image size.  OpenMVG is an empty submodule of the reference tree, so
// the two are restated from the textbook forms they implement (Hartley & Zisserman, Multiple View Geometry, 2nd ed.: algorithm 11.1 /
// section 11.1.2 for the seven-point pencil det(F1 + x F2) = 0, algorithm 4.1 for the DLT), with one stated difference:
//   the null space of the 7 x 9 (8 x 9) system is not read off an SVD (Eigen's JacobiSVD: an iteration with data-dependent sweeps) but
//   off a Householder QR of the transposed system -- the last 2 (1) columns of Q are an ORTHONORMAL basis of the same null space.  The
//   pencil, hence the set of solutions, does not depend on which orthonormal basis spans it; the QR is 7 (8) reflections with compile-
//   time loop bounds (no pivot search, nothing indexed at run time: it lives in registers on the GPU).
// Plain C++ for the host (tests/host/twoview_host_lib.cpp) and, through CLC_TV_HD, for gfx950; every file that includes it is built
// with contraction off, so an expression is the IEEE operations written here wherever it is inlined (cf. p3p.h).
#ifndef CLC_TWOVIEW_MIN_H
#define CLC_TWOVIEW_MIN_H

#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define CLC_TV_HD __host__ __device__ __forceinline__
#else
#define CLC_TV_HD static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define CLC_TV_UNROLL _Pragma("unroll")
#else
#define CLC_TV_UNROLL
#endif
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace clc {
namespace tv {

// Orthonormal basis Z of { z : A z = 0 } for the R x 9 matrix whose ROWS are B[0..R) (destroyed): Householder QR of A^T (9 x R), whose
// column k is row k of A; Z[j] = Q e_(R + j) = H_0 H_1 ... H_(R-1) e_(R + j).  A rank-deficient system (a degenerate sample) yields
// some basis of a subspace of the null space -- the models it gives are then judged by their residuals like any other.
template <int R>
CLC_TV_HD void nullspace(double (&B)[R][9], double (&Z)[9 - R][9])
{
    double beta[R];
    CLC_TV_UNROLL
    for (int k = 0; k < R; ++k) {
        double s = 0.0;
        CLC_TV_UNROLL
        for (int i = k; i < 9; ++i) s += B[k][i] * B[k][i];
        const double nx = sqrt(s);
        const double alpha = B[k][k] > 0.0 ? -nx : nx;
        B[k][k] -= alpha;                                            // v = x - alpha e_1, kept in place of x
        double vv = 0.0;
        CLC_TV_UNROLL
        for (int i = k; i < 9; ++i) vv += B[k][i] * B[k][i];
        beta[k] = vv > 0.0 ? 2.0 / vv : 0.0;
        CLC_TV_UNROLL
        for (int j = k + 1; j < R; ++j) {
            double d = 0.0;
            CLC_TV_UNROLL
            for (int i = k; i < 9; ++i) d += B[k][i] * B[j][i];
            d *= beta[k];
            CLC_TV_UNROLL
            for (int i = k; i < 9; ++i) B[j][i] -= d * B[k][i];
        }
    }
    CLC_TV_UNROLL
    for (int z = 0; z < 9 - R; ++z) {
        double e[9];
        CLC_TV_UNROLL
        for (int i = 0; i < 9; ++i) e[i] = i == R + z ? 1.0 : 0.0;
        CLC_TV_UNROLL
        for (int k = R - 1; k >= 0; --k) {
            double d = 0.0;
            CLC_TV_UNROLL
            for (int i = k; i < 9; ++i) d += B[k][i] * e[i];
            d *= beta[k];
            CLC_TV_UNROLL
            for (int i = k; i < 9; ++i) e[i] -= d * B[k][i];
        }
        CLC_TV_UNROLL
        for (int i = 0; i < 9; ++i) Z[z][i] = e[i];
    }
}

CLC_TV_HD double det3(const double* r0, const double* r1, const double* r2)
{
    return (r0[0] * (r1[1] * r2[2] - r1[2] * r2[1]) - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0])) + r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
}

// Real roots of x^3 + a x^2 + b x + c in the closed form of OpenMVG's SolveCubicPolynomial (numeric/poly.h, the GSL formulation):
// three roots in ascending order when the discriminant allows (a double root counted twice), one otherwise.
CLC_TV_HD int cubic_roots(const double a, const double b, const double c, double (&x)[3])
{
    const double q = a * a - 3.0 * b;
    const double r = (2.0 * a * a * a - 9.0 * a * b) + 27.0 * c;
    const double Q = q / 9.0, R = r / 54.0;
    const double Q3 = Q * Q * Q, R2 = R * R;
    const double CR2 = 729.0 * r * r, CQ3 = 2916.0 * q * q * q;
    const double a3 = a / 3.0;
    if (R == 0.0 && Q == 0.0) { x[0] = x[1] = x[2] = -a3; return 3; }
    if (CR2 == CQ3) {
        const double sq = sqrt(Q);
        if (R > 0.0) { x[0] = -2.0 * sq - a3; x[1] = sq - a3; x[2] = sq - a3; }
        else { x[0] = -sq - a3; x[1] = -sq - a3; x[2] = 2.0 * sq - a3; }
        return 3;
    }
    if (CR2 < CQ3) {
        const double sq = sqrt(Q);
        const double sq3 = sq * sq * sq;
        double cs = R / sq3;
        cs = cs > 1.0 ? 1.0 : (cs < -1.0 ? -1.0 : cs);
        const double theta = acos(cs);
        const double nrm = -2.0 * sq;
        double x0 = nrm * cos(theta / 3.0) - a3;
        double x1 = nrm * cos((theta + 2.0 * M_PI) / 3.0) - a3;
        double x2 = nrm * cos((theta - 2.0 * M_PI) / 3.0) - a3;
        double t;
        if (x0 > x1) { t = x0; x0 = x1; x1 = t; }
        if (x1 > x2) { t = x1; x1 = x2; x2 = t; if (x0 > x1) { t = x0; x0 = x1; x1 = t; } }
        x[0] = x0; x[1] = x1; x[2] = x2;
        return 3;
    }
    const double sgn = R >= 0.0 ? 1.0 : -1.0;
    const double A = -sgn * cbrt(fabs(R) + sqrt(R2 - Q3));
    const double Bq = Q / A;
    x[0] = (A + Bq) - a3;
    x[1] = x[2] = x[0];
    return 1;
}

// x1 / x2: the sample's 7 normalised correspondences {u, v}.  F[k] (row-major 9, x2^T F x1 = 0), k < the count returned (0..3).
CLC_TV_HD int seven_point(const double (&x1)[7][2], const double (&x2)[7][2], double (&F)[3][9])
{
    double A[7][9];
    CLC_TV_UNROLL
    for (int i = 0; i < 7; ++i) {
        const double u1 = x1[i][0], v1 = x1[i][1], u2 = x2[i][0], v2 = x2[i][1];
        A[i][0] = u2 * u1; A[i][1] = u2 * v1; A[i][2] = u2;
        A[i][3] = v2 * u1; A[i][4] = v2 * v1; A[i][5] = v2;
        A[i][6] = u1;      A[i][7] = v1;      A[i][8] = 1.0;
    }
    double Z[2][9];
    nullspace<7>(A, Z);
    const double* a = Z[0];
    const double* b = Z[1];
    // det(a + x b) = c0 + c1 x + c2 x^2 + c3 x^3: the determinant is linear in every row
    const double c0 = det3(a, a + 3, a + 6);
    const double c1 = (det3(b, a + 3, a + 6) + det3(a, b + 3, a + 6)) + det3(a, a + 3, b + 6);
    const double c2 = (det3(a, b + 3, b + 6) + det3(b, a + 3, b + 6)) + det3(b, b + 3, a + 6);
    const double c3 = det3(b, b + 3, b + 6);
    if (!(c3 != 0.0) || !(fabs(c3) <= 1.7976931348623157e308)) return 0;       // (a quadratic, or a NaN sample: no model)
    double roots[3];
    const int nr = cubic_roots(c2 / c3, c1 / c3, c0 / c3, roots);
    CLC_TV_UNROLL
    for (int k = 0; k < 3; ++k) {
        CLC_TV_UNROLL
        for (int e = 0; e < 9; ++e) F[k][e] = a[e] + roots[k] * b[e];
    }
    return nr;
}

// x1 / x2: the sample's 4 normalised correspondences.  H (row-major 9): x2 ~ H x1.  Always one model (FourPointSolver pushes one).
CLC_TV_HD int four_point(const double (&x1)[4][2], const double (&x2)[4][2], double (&H)[9])
{
    double L[8][9];
    CLC_TV_UNROLL
    for (int i = 0; i < 4; ++i) {
        const double u1 = x1[i][0], v1 = x1[i][1], u2 = x2[i][0], v2 = x2[i][1];
        double* r0 = L[2 * i];
        double* r1 = L[2 * i + 1];
        r0[0] = u1;  r0[1] = v1;  r0[2] = 1.0; r0[3] = 0.0; r0[4] = 0.0; r0[5] = 0.0; r0[6] = -u2 * u1; r0[7] = -u2 * v1; r0[8] = -u2;
        r1[0] = 0.0; r1[1] = 0.0; r1[2] = 0.0; r1[3] = u1;  r1[4] = v1;  r1[5] = 1.0; r1[6] = -v2 * u1; r1[7] = -v2 * v1; r1[8] = -v2;
    }
    double Z[1][9];
    nullspace<8>(L, Z);
    CLC_TV_UNROLL
    for (int e = 0; e < 9; ++e) H[e] = Z[0][e];
    return 1;
}

// ACKernelAdaptor's conditioning by the image size (openMVG/multiview/conditioning: PreconditionerFromPoints(width, height, T)):
// T = [ d 0 -w d / 2 ; 0 d -h d / 2 ; 0 0 1 ],  d = 1 / sqrt(w h);  x_n = T x
struct Normalizer { double d, tx, ty; };
CLC_TV_HD Normalizer normalizer(const int w, const int h)
{
    Normalizer t;
    t.d = 1.0 / sqrt((double)w * (double)h);
    t.tx = -0.5 * (double)w * t.d;
    t.ty = -0.5 * (double)h * t.d;
    return t;
}
CLC_TV_HD void mul3(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}
// UnnormalizerT: F = T2^T Fn T1 (fundamental);  UnnormalizerI: H = T2^-1 Hn T1 (homography).  Both images have the same size here
// (the reference hands params.imageSize for both, RobustMatcher.hpp:138-139, :199-201).
CLC_TV_HD void unnormalize(const bool homography, const Normalizer& t, const double* Mn, double* M)
{
    const double T1[9] = { t.d, 0.0, t.tx, 0.0, t.d, t.ty, 0.0, 0.0, 1.0 };
    const double T2t[9] = { t.d, 0.0, 0.0, 0.0, t.d, 0.0, t.tx, t.ty, 1.0 };
    const double T2i[9] = { 1.0 / t.d, 0.0, -t.tx / t.d, 0.0, 1.0 / t.d, -t.ty / t.d, 0.0, 0.0, 1.0 };
    double tmp[9];
    mul3(Mn, T1, tmp);
    mul3(homography ? T2i : T2t, tmp, M);
}

} // namespace tv
} // namespace clc
#endif
