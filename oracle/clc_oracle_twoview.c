/*
 * clc_oracle_twoview.c -- CPU ORACLE (test infrastructure, see clc_oracle.h) for the two other models RobustMatcher can
 * filter with (reference include/coloc/RobustMatcher.hpp:399-405):
 *   'F'  filterFundamental  :128-151   ACKernelAdaptor<SevenPointSolver, EpipolarDistanceError, UnnormalizerT>(x1, w, h, x2, w, h, true)
 *   'H'  filterHomography   :188-239   ACKernelAdaptor<FourPointSolver, AsymmetricError, UnnormalizerI>(x1, w, h, x2, w, h, false)
 *
 * PARITY UNPINNED, for the reason clc_oracle_acr.c gives: lib/openMVG is an empty submodule of the reference tree.  What is
 * restated here, from the textbook algorithms those OpenMVG classes implement (Hartley & Zisserman, 2nd ed., alg. 11.1 /
 * sec. 11.1.2 and alg. 4.1) and from OpenMVG's publicly documented structure:
 *   conditioning    x_n = T x,  T = [ d 0 -w d/2 ; 0 d -h d/2 ; 0 0 1 ],  d = 1 / sqrt(w h)
 *   seven points    rows [u2 u1, u2 v1, u2, v2 u1, v2 v1, v2, u1, v1, 1]; F1, F2 = the two right singular vectors of the
 *                   smallest singular values; real roots of det(F1 + x F2) = 0 (closed form, ascending); F = F1 + x F2
 *   four points     rows [u1 v1 1 0 0 0 -u2 u1 -u2 v1 -u2], [0 0 0 u1 v1 1 -v2 u1 -v2 v1 -v2]; H = the right singular vector of
 *                   the smallest singular value
 *   residuals       'F': (x2^T F x1)^2 / |(F x1)_xy|^2 (distance to the epipolar line in image 2, squared);
 *                   'H': |x2 - (H x1)_xy / (H x1)_z|^2   -- both on NORMALISED coordinates with the normalised model
 *   un-normalising  F = T^T Fn T;  H = T^-1 Hn T
 * The singular vectors come from a one-sided Jacobi SVD (the family Eigen's JacobiSVD belongs to); the product
 * (coloc_amd/csrc/twoview_min.h) takes an orthonormal null-space basis from a Householder QR instead -- a different basis of the
 * same pencil, so the two give the same matrices to rounding, root by root, but not bit for bit.  Shares no code with the product.
 */
#include "clc_oracle.h"

#include <math.h>
#include <string.h>

/* right singular vectors of the `rows` x 9 matrix A: V (9 x 9, column j in V[9 i + j]) and the singular values sv[9] (unsorted) */
static void jacobi_svd9(const double* A, int rows, double* V, double* sv)
{
    double G[9][9];
    memset(G, 0, sizeof G);
    for (int i = 0; i < rows; ++i) for (int j = 0; j < 9; ++j) G[i][j] = A[9 * i + j];
    for (int i = 0; i < 9; ++i) for (int j = 0; j < 9; ++j) V[9 * i + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = 0; i < 9; ++i) { al += G[i][p] * G[i][p]; be += G[i][q] * G[i][q]; ga += G[i][p] * G[i][q]; }
                if (ga == 0.0 || fabs(ga) <= 1e-16 * sqrt(al * be)) continue;
                rotated = 1;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 9; ++i) {
                    const double gp = G[i][p], gq = G[i][q];
                    G[i][p] = c * gp - s * gq;
                    G[i][q] = s * gp + c * gq;
                    const double vp = V[9 * i + p], vq = V[9 * i + q];
                    V[9 * i + p] = c * vp - s * vq;
                    V[9 * i + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < 9; ++j) {
        double s = 0.0;
        for (int i = 0; i < 9; ++i) s += G[i][j] * G[i][j];
        sv[j] = sqrt(s);
    }
}

/* the `want` right singular vectors of the smallest singular values, smallest first: Z[k][0..9) */
static void smallest_vectors(const double* A, int rows, int want, double (*Z)[9])
{
    double V[81], sv[9];
    int used[9] = { 0 };
    jacobi_svd9(A, rows, V, sv);
    for (int k = 0; k < want; ++k) {
        int best = -1;
        for (int j = 0; j < 9; ++j) if (!used[j] && (best < 0 || sv[j] < sv[best])) best = j;
        used[best] = 1;
        for (int i = 0; i < 9; ++i) Z[k][i] = V[9 * i + best];
    }
}

static double det_rows(const double* r0, const double* r1, const double* r2)
{
    return r0[0] * r1[1] * r2[2] + r0[1] * r1[2] * r2[0] + r0[2] * r1[0] * r2[1]
         - r0[2] * r1[1] * r2[0] - r0[1] * r1[0] * r2[2] - r0[0] * r1[2] * r2[1];
}

/* real roots of p3 x^3 + p2 x^2 + p1 x + p0, ascending (one root: 1) */
static int cubic_real_roots(double p0, double p1, double p2, double p3, double* x)
{
    if (p3 == 0.0) return 0;
    const double a = p2 / p3, b = p1 / p3, c = p0 / p3;
    const double q = a * a - 3 * b, r = 2 * a * a * a - 9 * a * b + 27 * c;
    const double Q = q / 9, R = r / 54;
    const double Q3 = Q * Q * Q, R2 = R * R;
    const double CR2 = 729 * r * r, CQ3 = 2916 * q * q * q;
    if (R == 0 && Q == 0) { x[0] = x[1] = x[2] = -a / 3; return 3; }
    if (CR2 == CQ3) {
        const double sq = sqrt(Q);
        if (R > 0) { x[0] = -2 * sq - a / 3; x[1] = sq - a / 3; x[2] = sq - a / 3; }
        else { x[0] = -sq - a / 3; x[1] = -sq - a / 3; x[2] = 2 * sq - a / 3; }
        return 3;
    }
    if (CR2 < CQ3) {
        const double sq = sqrt(Q), sq3 = sq * sq * sq;
        double cs = R / sq3;
        if (cs > 1) cs = 1;
        if (cs < -1) cs = -1;
        const double theta = acos(cs), nrm = -2 * sq;
        double v[3] = { nrm * cos(theta / 3) - a / 3, nrm * cos((theta + 2.0 * M_PI) / 3) - a / 3, nrm * cos((theta - 2.0 * M_PI) / 3) - a / 3 };
        for (int i = 0; i < 3; ++i) for (int j = i + 1; j < 3; ++j) if (v[j] < v[i]) { const double t = v[i]; v[i] = v[j]; v[j] = t; }
        x[0] = v[0]; x[1] = v[1]; x[2] = v[2];
        return 3;
    }
    const double sgn = R >= 0 ? 1 : -1;
    const double A = -sgn * pow(fabs(R) + sqrt(R2 - Q3), 1.0 / 3.0);
    x[0] = A + Q / A - a / 3;
    return 1;
}

int orc_seven_point(const double* q1, const double* q2, double* F_out)
{
    double A[7 * 9];
    for (int i = 0; i < 7; ++i) {
        const double u1 = q1[2 * i], v1 = q1[2 * i + 1], u2 = q2[2 * i], v2 = q2[2 * i + 1];
        double* r = A + 9 * i;
        r[0] = u2 * u1; r[1] = u2 * v1; r[2] = u2; r[3] = v2 * u1; r[4] = v2 * v1; r[5] = v2; r[6] = u1; r[7] = v1; r[8] = 1.0;
    }
    double Z[2][9];
    smallest_vectors(A, 7, 2, Z);
    const double *f1 = Z[0], *f2 = Z[1];
    /* det(F1 + x F2): the determinant is multilinear in the rows */
    const double p0 = det_rows(f1, f1 + 3, f1 + 6);
    const double p1 = det_rows(f2, f1 + 3, f1 + 6) + det_rows(f1, f2 + 3, f1 + 6) + det_rows(f1, f1 + 3, f2 + 6);
    const double p2 = det_rows(f1, f2 + 3, f2 + 6) + det_rows(f2, f1 + 3, f2 + 6) + det_rows(f2, f2 + 3, f1 + 6);
    const double p3 = det_rows(f2, f2 + 3, f2 + 6);
    double roots[3];
    const int n = cubic_real_roots(p0, p1, p2, p3, roots);
    for (int k = 0; k < n; ++k) for (int e = 0; e < 9; ++e) F_out[9 * k + e] = f1[e] + roots[k] * f2[e];
    return n;
}

int orc_four_point(const double* q1, const double* q2, double* H_out)
{
    double L[8 * 9];
    memset(L, 0, sizeof L);
    for (int i = 0; i < 4; ++i) {
        const double u1 = q1[2 * i], v1 = q1[2 * i + 1], u2 = q2[2 * i], v2 = q2[2 * i + 1];
        double* r = L + 18 * i;
        r[0] = u1; r[1] = v1; r[2] = 1.0; r[6] = -u2 * u1; r[7] = -u2 * v1; r[8] = -u2;
        r += 9;
        r[3] = u1; r[4] = v1; r[5] = 1.0; r[6] = -v2 * u1; r[7] = -v2 * v1; r[8] = -v2;
    }
    double Z[1][9];
    smallest_vectors(L, 8, 1, Z);
    memcpy(H_out, Z[0], sizeof(double) * 9);
    return 1;
}

void orc_tv_normalizer(int w, int h, double* d_tx_ty)
{
    const double d = 1.0 / sqrt((double)w * (double)h);
    d_tx_ty[0] = d;
    d_tx_ty[1] = -0.5 * (double)w * d;
    d_tx_ty[2] = -0.5 * (double)h * d;
}

void orc_tv_normalize(int w, int h, const double* x, int n, double* xn)
{
    double t[3];
    orc_tv_normalizer(w, h, t);
    for (int i = 0; i < n; ++i) { xn[2 * i] = x[2 * i] * t[0] + t[1]; xn[2 * i + 1] = x[2 * i + 1] * t[0] + t[2]; }
}

static void mat3_mul(const double* A, const double* B, double* C)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}

void orc_tv_unnormalize(int homography, int w, int h, const double* Mn, double* M)
{
    double t[3], tmp[9];
    orc_tv_normalizer(w, h, t);
    const double T[9] = { t[0], 0, t[1], 0, t[0], t[2], 0, 0, 1 };
    const double Tt[9] = { t[0], 0, 0, 0, t[0], 0, t[1], t[2], 1 };
    const double Ti[9] = { 1.0 / t[0], 0, -t[1] / t[0], 0, 1.0 / t[0], -t[2] / t[0], 0, 0, 1 };
    mat3_mul(Mn, T, tmp);
    mat3_mul(homography ? Ti : Tt, tmp, M);
}

/* EpipolarDistanceError (kind 2) / AsymmetricError (kind 3) of one model over n normalised correspondences */
void orc_tv_residuals(int kind, const double* M, const double* q1, const double* q2, int n, double* e)
{
    for (int i = 0; i < n; ++i) {
        const double u1 = q1[2 * i], v1 = q1[2 * i + 1], u2 = q2[2 * i], v2 = q2[2 * i + 1];
        const double a0 = (M[0] * u1 + M[1] * v1) + M[2];
        const double a1 = (M[3] * u1 + M[4] * v1) + M[5];
        const double a2 = (M[6] * u1 + M[7] * v1) + M[8];
        if (kind == 2) {
            const double d = (u2 * a0 + v2 * a1) + a2;
            e[i] = (d * d) / (a0 * a0 + a1 * a1);
        } else {
            const double du = u2 - a0 / a2, dv = v2 - a1 / a2;
            e[i] = du * du + dv * dv;
        }
    }
}
