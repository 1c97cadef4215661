// fivept.h -- calibrated five-point relative pose: up to 10 essential matrices from 5 correspondences, one
// problem per lane (host + device inline).
//
// Role: hypothesis generator for the batched essential-matrix RANSAC (clc_essential_ransac), the GPU form of
// RobustMatcher::filterEssential (reference include/coloc/RobustMatcher.hpp:153-186), which asks OpenMVG for
// essential::kernel::FivePointSolver inside AC-RANSAC.  OpenMVG is an empty, unpinned submodule in the
// reference snapshot, so this follows the published method (Nister 2004 / Stewenius et al. 2006), derived
// from scratch and built by polynomial arithmetic rather than by a transcribed coefficient table:
//   1. the 5 epipolar constraints q2^T E q1 = 0 leave a 4-D null space {E1..E4}: E = x E1 + y E2 + z E3 + E4;
//   2. det(E) = 0 and 2 E E^T E - trace(E E^T) E = 0 are 10 cubic polynomials in (x, y, z) over 20 monomials,
//      ordered [x3 x2y x2z xy2 xyz xz2 y3 y2z yz2 z3 | x2 xy xz y2 yz z2 x y z 1];
//   3. Gauss-Jordan on the 10 cubic monomials gives [I | B]; with basis b = [x2 xy xz y2 yz z2 x y z 1]^T the
//      action matrix of "multiply by x" has rows -B[x3], -B[x2y], -B[x2z], -B[xy2], -B[xyz], -B[xz2] and the
//      unit rows x*x = x2, x*y = xy, x*z = xz, x*1 = x;  A b = x b at every solution;
//   4. eigenvalues by Hessenberg reduction + real double-shift QR; each real one is refined together with its
//      eigenvector by inverse iteration on A itself;  (x, y, z) = b[6..8] / b[9].
//      (A characteristic-polynomial route -- Faddeev-LeVerrier + Durand-Kerner -- lost 15 % of the true solutions
//      to cancellation in the coefficients and was dropped.)
// fp64 throughout.  Inputs are NORMALISED image coordinates (K^-1 applied).
#ifndef CLC_FIVEPT_H
#define CLC_FIVEPT_H

#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define FPT_HD __host__ __device__ inline
#else
#define FPT_HD static inline
#endif

// Every array of the solver.  On the device ONE problem runs per wave (all lanes execute the same scalar code on
// the same numbers), and this struct lives in LDS: kept as per-lane locals the arrays would be dynamically indexed
// private (scratch) memory, and a solve is a chain of ~10^5 dependent accesses (measured 4.4 ms per problem in
// scratch).  On the host it is a local variable.
struct FptWorkspace {
    double A[5][9], EE[4][9], e[9][20], M[10][20], G[9][20], tr[20], t[20], M0[10][20], Ax[100];
    double hr[10][10];                                            // eigenvalues: the Hessenberg matrix
    double zr[10], zi[10];
    int piv[5], is_piv[9];
    // one slot per eigenvalue: on the device the ten roots are refined by ten lanes at once
    struct Root {
        double lu[10][11];                                        // shifted solve
        double v[10], y[10];
        double cand[9];                                           // the essential matrix this root gives
        int valid;
    } root[10];
};

// exponents of the 20 monomials in the order above
FPT_HD int fpt_mono_index(int i, int j, int k)
{
    // i, j, k = exponents of x, y, z; total degree <= 3
    const int d = i + j + k;
    if (d == 3) {
        if (i == 3) return 0;
        if (i == 2) return j == 1 ? 1 : 2;
        if (i == 1) return j == 2 ? 3 : (j == 1 ? 4 : 5);
        return j == 3 ? 6 : (j == 2 ? 7 : (j == 1 ? 8 : 9));
    }
    if (d == 2) {
        if (i == 2) return 10;
        if (i == 1) return j == 1 ? 11 : 12;
        return j == 2 ? 13 : (j == 1 ? 14 : 15);
    }
    if (d == 1) return i == 1 ? 16 : (j == 1 ? 17 : 18);
    return 19;
}
FPT_HD void fpt_mono_exp(int m, int* i, int* j, int* k)
{
    const signed char E[20][3] = { {3,0,0},{2,1,0},{2,0,1},{1,2,0},{1,1,1},{1,0,2},{0,3,0},{0,2,1},{0,1,2},{0,0,3},
                                   {2,0,0},{1,1,0},{1,0,1},{0,2,0},{0,1,1},{0,0,2},{1,0,0},{0,1,0},{0,0,1},{0,0,0} };
    *i = E[m][0]; *j = E[m][1]; *k = E[m][2];
}

// c += a * b for polynomials stored over the 20 monomials (caller guarantees deg(a) + deg(b) <= 3)
FPT_HD void fpt_poly_mul_add(const double* a, const double* b, double scale, double* c)
{
    for (int m = 0; m < 20; ++m) {
        if (a[m] == 0.0) continue;
        int ai, aj, ak;
        fpt_mono_exp(m, &ai, &aj, &ak);
        for (int n = 0; n < 20; ++n) {
            if (b[n] == 0.0) continue;
            int bi, bj, bk;
            fpt_mono_exp(n, &bi, &bj, &bk);
            if (ai + aj + ak + bi + bj + bk > 3) continue;
            c[fpt_mono_index(ai + bi, aj + bj, ak + bk)] += scale * a[m] * b[n];
        }
    }
}

// ---- structured polynomial products -------------------------------------------------------------------------
// The entries of E are LINEAR in (x, y, z) (monomials 16..19), the entries of E E^T QUADRATIC (monomials 10..19):
// the only products the constraints need are linear x linear (16 terms) and quadratic x linear (40 terms), with the
// target monomial of every term known at compile time.  (The generic 20 x 20 loop with exponent look-ups that this
// replaces was half of the solver's run time.)
constexpr signed char kFptExp[20][3] = { {3,0,0},{2,1,0},{2,0,1},{1,2,0},{1,1,1},{1,0,2},{0,3,0},{0,2,1},{0,1,2},{0,0,3},
                                         {2,0,0},{1,1,0},{1,0,1},{0,2,0},{0,1,1},{0,0,2},{1,0,0},{0,1,0},{0,0,1},{0,0,0} };
constexpr int fpt_mono_index_c(int i, int j, int k)
{
    return (i + j + k == 3) ? (i == 3 ? 0 : i == 2 ? (j == 1 ? 1 : 2) : i == 1 ? (j == 2 ? 3 : (j == 1 ? 4 : 5)) : (j == 3 ? 6 : (j == 2 ? 7 : (j == 1 ? 8 : 9))))
         : (i + j + k == 2) ? (i == 2 ? 10 : i == 1 ? (j == 1 ? 11 : 12) : (j == 2 ? 13 : (j == 1 ? 14 : 15)))
         : (i + j + k == 1) ? (i == 1 ? 16 : (j == 1 ? 17 : 18)) : 19;
}
constexpr int fpt_prod_index_c(int m, int n)
{
    return fpt_mono_index_c(kFptExp[m][0] + kFptExp[n][0], kFptExp[m][1] + kFptExp[n][1], kFptExp[m][2] + kFptExp[n][2]);
}
// c += scale * a * b, a and b linear (coefficients at 16..19), c quadratic
FPT_HD void fpt_mul_lin_lin(const double* a, const double* b, double scale, double* c)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int m = 16; m < 20; ++m) {
        const double am = scale * a[m];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int n = 16; n < 20; ++n) c[fpt_prod_index_c(m, n)] += am * b[n];
    }
}
// c += scale * a * b, a quadratic (coefficients at 10..19), b linear, c cubic
FPT_HD void fpt_mul_quad_lin(const double* a, const double* b, double scale, double* c)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int m = 10; m < 20; ++m) {
        const double am = scale * a[m];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int n = 16; n < 20; ++n) c[fpt_prod_index_c(m, n)] += am * b[n];
    }
}

// solve (A - lambda I) y = rhs for a 10 x 10 A by LU with partial pivoting; returns false if singular to working precision
FPT_HD bool fpt_solve_shifted(const double* A, double lambda, const double* rhs, double* y, double (&M)[10][11])
{
    for (int r = 0; r < 10; ++r) {
        for (int c = 0; c < 10; ++c) M[r][c] = A[10 * r + c] - (r == c ? lambda : 0.0);
        M[r][10] = rhs[r];
    }
    for (int c = 0; c < 10; ++c) {
        int p = c;
        double best = fabs(M[c][c]);
        for (int r = c + 1; r < 10; ++r) if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); p = r; }
        if (!(best > 0.0)) { M[c][c] = 1e-300; best = 1e-300; p = c; }
        if (p != c) for (int k = 0; k < 11; ++k) { const double t = M[c][k]; M[c][k] = M[p][k]; M[p][k] = t; }
        const double inv = 1.0 / M[c][c];
        for (int r = c + 1; r < 10; ++r) {
            const double f = M[r][c] * inv;
            if (f == 0.0) continue;
            for (int k = c; k < 11; ++k) M[r][k] -= f * M[c][k];
        }
    }
    for (int r = 9; r >= 0; --r) {
        double s = M[r][10];
        for (int k = r + 1; k < 10; ++k) s -= M[r][k] * y[k];
        y[r] = s / M[r][r];
    }
    for (int r = 0; r < 10; ++r) if (!(y[r] == y[r])) return false;
    return true;
}

// Element loops of the eigenvalue routine: on the device the wave that owns the problem spreads them over its lanes
// (every element update of a row / column operation is independent), on the host they are ordinary loops.  The
// arithmetic per element is the same either way.
#if defined(__HIP_DEVICE_COMPILE__)
#define FPT_PAR_FOR(j, lo, hi) for (int j = (lo) + (int)threadIdx.x, once_ = 1; once_ && j <= (hi); once_ = 0)
#define FPT_SYNC() __syncthreads()
#else
#define FPT_PAR_FOR(j, lo, hi) for (int j = (lo); j <= (hi); ++j)
#define FPT_SYNC() ((void)0)
#endif

// All eigenvalues of a real 10 x 10 matrix (copied; A is not modified): Hessenberg form by stabilised elimination,
// then the real double-shift QR iteration (Francis steps with 3-element Householder reflections, the classic EISPACK
// "hqr" scheme) -- no complex arithmetic, a conjugate pair converges in one go.  (A complex single-shift QR gave the
// same hit rate and took 8 % longer end to end.)
FPT_HD void fpt_eigenvalues10(const double* A, double* wr, double* wi, FptWorkspace& w)
{
    const int n = 10;
    double (&a)[10][10] = w.hr;
    FPT_SYNC();
    FPT_PAR_FOR(idx, 0, 99) a[idx / 10][idx % 10] = A[idx];
#if defined(__HIP_DEVICE_COMPILE__)
    FPT_PAR_FOR(idx, 64, 99) a[idx / 10][idx % 10] = A[idx];
#endif
    FPT_SYNC();
    // Hessenberg by elimination with row/column interchanges (similarity transforms)
    for (int m = 1; m < n - 1; ++m) {
        int p = m;
        double x = 0.0;
        for (int j = m; j < n; ++j) if (fabs(a[j][m - 1]) > fabs(x)) { x = a[j][m - 1]; p = j; }
        FPT_SYNC();
        if (p != m) {
            FPT_PAR_FOR(j, m - 1, n - 1) { const double t = a[p][j]; a[p][j] = a[m][j]; a[m][j] = t; }
            FPT_SYNC();
            FPT_PAR_FOR(j, 0, n - 1) { const double t = a[j][p]; a[j][p] = a[j][m]; a[j][m] = t; }
            FPT_SYNC();
        }
        if (x != 0.0) {
            for (int i = m + 1; i < n; ++i) {
                double y = a[i][m - 1];
                FPT_SYNC();
                if (y == 0.0) continue;
                y /= x;
                FPT_PAR_FOR(j, m, n - 1) a[i][j] -= y * a[m][j];
                FPT_PAR_FOR(j, m - 1, m - 1) a[i][j] = 0.0;
                FPT_SYNC();
                FPT_PAR_FOR(j, 0, n - 1) a[j][m] += y * a[j][i];
                FPT_SYNC();
            }
        }
    }
    FPT_SYNC();
    for (int r = 2; r < n; ++r) FPT_PAR_FOR(c, 0, r - 2) a[r][c] = 0.0;
    FPT_SYNC();
    double anorm = 0.0;
    for (int r = 0; r < n; ++r) for (int c = (r > 0 ? r - 1 : 0); c < n; ++c) anorm += fabs(a[r][c]);
    if (!(anorm > 0.0)) { FPT_SYNC(); FPT_PAR_FOR(i, 0, n - 1) { wr[i] = 0.0; wi[i] = 0.0; } FPT_SYNC(); return; }
    int nn = n - 1;
    double t = 0.0;
    while (nn >= 0) {
        int its = 0, l;
        do {
            for (l = nn; l >= 1; --l) {
                double s = fabs(a[l - 1][l - 1]) + fabs(a[l][l]);
                if (s == 0.0) s = anorm;
                if (fabs(a[l][l - 1]) <= 1e-13 * s) break;
            }
            FPT_SYNC();
            if (l >= 1) { FPT_PAR_FOR(j, 0, 0) a[l][l - 1] = 0.0; }
            FPT_SYNC();
            double x = a[nn][nn];
            if (l == nn) {                                   // one root
                FPT_SYNC();
                FPT_PAR_FOR(j, 0, 0) { wr[nn] = x + t; wi[nn] = 0.0; }
                --nn;
            } else {
                double y = a[nn - 1][nn - 1], ww = a[nn][nn - 1] * a[nn - 1][nn];
                if (l == nn - 1) {                           // two roots
                    const double p = 0.5 * (y - x), q = p * p + ww;
                    double z = sqrt(fabs(q));
                    x += t;
                    double r0, r1, i0, i1;
                    if (q >= 0.0) {
                        z = p + (p >= 0.0 ? fabs(z) : -fabs(z));
                        r0 = r1 = x + z;
                        if (z != 0.0) r1 = x - ww / z;
                        i0 = i1 = 0.0;
                    } else {
                        r0 = r1 = x + p;
                        i0 = -z; i1 = z;
                    }
                    FPT_SYNC();
                    FPT_PAR_FOR(j, 0, 0) { wr[nn - 1] = r0; wr[nn] = r1; wi[nn - 1] = i0; wi[nn] = i1; }
                    nn -= 2;
                } else {                                     // no roots yet: one more double step
                    if (its == 60) {                         // give up on this block: report the diagonal
                        FPT_SYNC();
                        FPT_PAR_FOR(i, l, nn) { wr[i] = a[i][i] + t; wi[i] = 0.0; }
                        nn = l - 1;
                        break;
                    }
                    if (its == 10 || its == 20 || its == 30 || its == 40 || its == 50) {     // exceptional shift
                        t += x;
                        FPT_SYNC();
                        FPT_PAR_FOR(i, 0, nn) a[i][i] -= x;
                        FPT_SYNC();
                        const double s = fabs(a[nn][nn - 1]) + fabs(a[nn - 1][nn - 2]);
                        y = x = 0.75 * s;
                        ww = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    double p = 0.0, q = 0.0, r = 0.0, z;
                    for (m = nn - 2; m >= l; --m) {          // shift + look for two consecutive small subdiagonal elements
                        z = a[m][m];
                        r = x - z;
                        double s = y - z;
                        p = (r * s - ww) / a[m + 1][m] + a[m][m + 1];
                        q = a[m + 1][m + 1] - z - r - s;
                        r = a[m + 2][m + 1];
                        s = fabs(p) + fabs(q) + fabs(r);
                        if (s != 0.0) { p /= s; q /= s; r /= s; }
                        if (m == l) break;
                        const double u = fabs(a[m][m - 1]) * (fabs(q) + fabs(r));
                        const double v = fabs(p) * (fabs(a[m - 1][m - 1]) + fabs(z) + fabs(a[m + 1][m + 1]));
                        if (u <= 1e-16 * v) break;
                    }
                    FPT_SYNC();
                    FPT_PAR_FOR(i, m + 2, nn) { a[i][i - 2] = 0.0; if (i != m + 2) a[i][i - 3] = 0.0; }
                    FPT_SYNC();
                    for (int k = m; k <= nn - 1; ++k) {      // double QR step on rows l..nn and columns m..nn
                        if (k != m) {
                            p = a[k][k - 1];
                            q = a[k + 1][k - 1];
                            r = k != nn - 1 ? a[k + 2][k - 1] : 0.0;
                            x = fabs(p) + fabs(q) + fabs(r);
                            if (x != 0.0) { p /= x; q /= x; r /= x; }
                        }
                        const double sn = sqrt(p * p + q * q + r * r);
                        const double s = p >= 0.0 ? sn : -sn;
                        FPT_SYNC();
                        if (s != 0.0) {
                            if (k == m) { if (l != m) { FPT_PAR_FOR(j, 0, 0) a[k][k - 1] = -a[k][k - 1]; } }
                            else { FPT_PAR_FOR(j, 0, 0) a[k][k - 1] = -s * x; }
                            p += s;
                            x = p / s; y = q / s; z = r / s;
                            q /= p; r /= p;
                            const bool three = k != nn - 1;
                            FPT_SYNC();
                            FPT_PAR_FOR(j, k, nn) {                       // row modification
                                double pp = a[k][j] + q * a[k + 1][j];
                                if (three) { pp += r * a[k + 2][j]; a[k + 2][j] -= pp * z; }
                                a[k + 1][j] -= pp * y;
                                a[k][j] -= pp * x;
                            }
                            FPT_SYNC();
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            FPT_PAR_FOR(i, l, mmin) {                     // column modification
                                double pp = x * a[i][k] + y * a[i][k + 1];
                                if (three) { pp += z * a[i][k + 2]; a[i][k + 2] -= pp * r; }
                                a[i][k + 1] -= pp * q;
                                a[i][k] -= pp;
                            }
                            FPT_SYNC();
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    FPT_SYNC();
}

// relative residual of the essential-matrix constraints 2 E E^T E - tr(E E^T) E = 0 (which imply det E = 0)
FPT_HD double fpt_constraint_residual(const double* E)
{
    double G[9], n2 = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) G[3 * a + b] = E[3 * a] * E[3 * b] + E[3 * a + 1] * E[3 * b + 1] + E[3 * a + 2] * E[3 * b + 2];
    for (int i = 0; i < 9; ++i) n2 += E[i] * E[i];
    const double tr = G[0] + G[4] + G[8];
    double worst = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const double c = 2.0 * (G[3 * a] * E[b] + G[3 * a + 1] * E[3 + b] + G[3 * a + 2] * E[6 + b]) - tr * E[3 * a + b];
            worst = fabs(c) > worst ? fabs(c) : worst;
        }
    return n2 > 0.0 ? worst / (n2 * sqrt(n2)) : 1.0;
}

// Eigenvalue k of the action matrix -> candidate essential matrix w.root[k].cand (valid = 0 if the eigenvalue is
// complex, the eigenvector does not settle, or the result does not satisfy the cubic constraints).
FPT_HD void fpt_root_candidate(FptWorkspace& w, const int k, const double anorm)
{
    FptWorkspace::Root& rw = w.root[k];
    rw.valid = 0;
    const double (&Ax)[100] = w.Ax;
    const double (&zr)[10] = w.zr;
    const double (&zi)[10] = w.zi;
    const double (&M0)[10][20] = w.M0;
    const double (&EE)[4][9] = w.EE;
    {
        if (fabs(zi[k]) > 1e-6 * (1.0 + fabs(zr[k]))) return;
        double lam = zr[k];
        double (&v)[10] = rw.v;
        double (&y)[10] = rw.y;
        for (int i = 0; i < 10; ++i) v[i] = 1.0 / (1.0 + i);
        bool ok = true;
        for (int it = 0; it < 2 && ok; ++it) {          // two steps settle it (the polish below does the rest)
            ok = fpt_solve_shifted(Ax, lam + 1e-11 * (anorm + 1.0), v, y, rw.lu);
            if (!ok) break;
            double nrm = 0.0;
            for (int i = 0; i < 10; ++i) nrm = fabs(y[i]) > nrm ? fabs(y[i]) : nrm;
            if (!(nrm > 0.0)) { ok = false; break; }
            for (int i = 0; i < 10; ++i) v[i] = y[i] / nrm;
            // eigenvalue from the component of largest magnitude: (A v)_m / v_m
            int mi = 0;
            for (int i = 1; i < 10; ++i) if (fabs(v[i]) > fabs(v[mi])) mi = i;
            double av = 0.0;
            for (int j = 0; j < 10; ++j) av += Ax[10 * mi + j] * v[j];
            lam = av / v[mi];
        }
        if (!ok || fabs(v[9]) < 1e-12) return;
        double x = v[6] / v[9], yv = v[7] / v[9], z = v[8] / v[9];
        // polish (x, y, z) on the ten cubic constraints themselves (Gauss-Newton, 3 unknowns): the eigen-solution
        // of a nearly defective action matrix is only good to ~1e-5, the constraints pin it to rounding
        for (int it = 0; it < 4; ++it) {
            // monomials and their derivatives with compile-time exponents, loops fully unrolled on the device: the 80
            // values stay in registers and the 10 x 20 x 4 multiply-adds below read M0 at fixed LDS offsets
            double mono[20], dmx[20], dmy[20], dmz[20];
            const double px[4] = { 1.0, x, x * x, x * x * x }, py[4] = { 1.0, yv, yv * yv, yv * yv * yv }, pz[4] = { 1.0, z, z * z, z * z * z };
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int m = 0; m < 20; ++m) {
                const int ei = kFptExp[m][0], ej = kFptExp[m][1], ek = kFptExp[m][2];
                mono[m] = px[ei] * py[ej] * pz[ek];
                dmx[m] = ei ? ei * px[ei ? ei - 1 : 0] * py[ej] * pz[ek] : 0.0;
                dmy[m] = ej ? ej * px[ei] * py[ej ? ej - 1 : 0] * pz[ek] : 0.0;
                dmz[m] = ek ? ek * px[ei] * py[ej] * pz[ek ? ek - 1 : 0] : 0.0;
            }
            double JtJ[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, Jtr[3] = { 0, 0, 0 };
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int r = 0; r < 10; ++r) {
                double rr = 0.0, jx = 0.0, jy = 0.0, jz = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int m = 0; m < 20; ++m) { rr += M0[r][m] * mono[m]; jx += M0[r][m] * dmx[m]; jy += M0[r][m] * dmy[m]; jz += M0[r][m] * dmz[m]; }
                JtJ[0] += jx * jx; JtJ[1] += jx * jy; JtJ[2] += jx * jz; JtJ[4] += jy * jy; JtJ[5] += jy * jz; JtJ[8] += jz * jz;
                Jtr[0] += jx * rr; Jtr[1] += jy * rr; Jtr[2] += jz * rr;
            }
            JtJ[3] = JtJ[1]; JtJ[6] = JtJ[2]; JtJ[7] = JtJ[5];
            const double det = JtJ[0] * (JtJ[4] * JtJ[8] - JtJ[5] * JtJ[7]) - JtJ[1] * (JtJ[3] * JtJ[8] - JtJ[5] * JtJ[6])
                             + JtJ[2] * (JtJ[3] * JtJ[7] - JtJ[4] * JtJ[6]);
            if (!(fabs(det) > 1e-300)) break;
            const double dx_ = (Jtr[0] * (JtJ[4] * JtJ[8] - JtJ[5] * JtJ[7]) - JtJ[1] * (Jtr[1] * JtJ[8] - JtJ[5] * Jtr[2])
                              + JtJ[2] * (Jtr[1] * JtJ[7] - JtJ[4] * Jtr[2])) / det;
            const double dy_ = (JtJ[0] * (Jtr[1] * JtJ[8] - JtJ[5] * Jtr[2]) - Jtr[0] * (JtJ[3] * JtJ[8] - JtJ[5] * JtJ[6])
                              + JtJ[2] * (JtJ[3] * Jtr[2] - Jtr[1] * JtJ[6])) / det;
            const double dz_ = (JtJ[0] * (JtJ[4] * Jtr[2] - Jtr[1] * JtJ[7]) - JtJ[1] * (JtJ[3] * Jtr[2] - Jtr[1] * JtJ[6])
                              + Jtr[0] * (JtJ[3] * JtJ[7] - JtJ[4] * JtJ[6])) / det;
            if (!(dx_ == dx_) || !(dy_ == dy_) || !(dz_ == dz_)) break;
            x -= dx_; yv -= dy_; z -= dz_;
            if (fabs(dx_) + fabs(dy_) + fabs(dz_) < 1e-15 * (1.0 + fabs(x) + fabs(yv) + fabs(z))) break;
        }
        bool finite = true;
        for (int c = 0; c < 9; ++c) {
            const double ev = x * EE[0][c] + yv * EE[1][c] + z * EE[2][c] + EE[3][c];
            rw.cand[c] = ev;
            finite = finite && (ev == ev) && fabs(ev) < 1e300;
        }
        // an eigenvalue that was not a true real root (or an eigenvector that did not settle) yields a matrix that
        // is not essential: keep only candidates that satisfy the cubic constraints
        if (finite && fpt_constraint_residual(rw.cand) < 1e-9) rw.valid = 1;
    }
}

// q1, q2: 5 x 2 normalised coordinates in view 1 / view 2 (q2^T E q1 = 0).  E_out: up to 10 x 9 (row-major 3x3).
FPT_HD int fivept_solve(const double q1[5][2], const double q2[5][2], double* E_out, FptWorkspace& w)
{
    // ---- 1. null space of the 5 x 9 constraint matrix by reduced row echelon form
    double (&A)[5][9] = w.A;
    for (int i = 0; i < 5; ++i) {
        const double a[3] = { q2[i][0], q2[i][1], 1.0 }, b[3] = { q1[i][0], q1[i][1], 1.0 };
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[i][3 * r + c] = a[r] * b[c];
    }
    int (&piv)[5] = w.piv;
    int (&is_piv)[9] = w.is_piv;
    for (int i = 0; i < 9; ++i) is_piv[i] = 0;
    int row = 0;
    for (int col = 0; col < 9 && row < 5; ++col) {
        int p = row;
        double best = fabs(A[row][col]);
        for (int r = row + 1; r < 5; ++r) if (fabs(A[r][col]) > best) { best = fabs(A[r][col]); p = r; }
        if (best < 1e-12) continue;
        if (p != row) for (int k = 0; k < 9; ++k) { const double t = A[row][k]; A[row][k] = A[p][k]; A[p][k] = t; }
        const double inv = 1.0 / A[row][col];
        for (int k = 0; k < 9; ++k) A[row][k] *= inv;
        for (int r = 0; r < 5; ++r) {
            if (r == row) continue;
            const double f = A[r][col];
            if (f == 0.0) continue;
            for (int k = 0; k < 9; ++k) A[r][k] -= f * A[row][k];
        }
        piv[row] = col;
        is_piv[col] = 1;
        ++row;
    }
    if (row < 5) return 0;                      // degenerate sample
    double (&EE)[4][9] = w.EE;
    {
        int nb = 0;
        for (int f = 0; f < 9; ++f) {
            if (is_piv[f]) continue;
            for (int k = 0; k < 9; ++k) EE[nb][k] = 0.0;
            EE[nb][f] = 1.0;
            for (int r = 0; r < 5; ++r) EE[nb][piv[r]] = -A[r][f];
            ++nb;
        }
    }
    // ---- 2. the ten cubic constraints.  e[a][b] = linear polynomial x EE0 + y EE1 + z EE2 + EE3
    double (&e)[9][20] = w.e;
    for (int k = 0; k < 9; ++k) {
        for (int m = 0; m < 20; ++m) e[k][m] = 0.0;
        e[k][16] = EE[0][k]; e[k][17] = EE[1][k]; e[k][18] = EE[2][k]; e[k][19] = EE[3][k];
    }
    double (&M)[10][20] = w.M;
    double (&G)[9][20] = w.G;
    double (&tr)[20] = w.tr;
    // (on the device: one polynomial per lane -- ten lanes build G = E E^T and det(E), then nine lanes the rows of
    //  2 G E - tr(G) E; on the host the same in turn)
    FPT_SYNC();
    FPT_PAR_FOR(g, 0, 9) {
        if (g < 9) {
            // G = E E^T (quadratic)
            const int a = g / 3, b = g - 3 * a;
            for (int m = 0; m < 20; ++m) G[g][m] = 0.0;
            for (int c = 0; c < 3; ++c) fpt_mul_lin_lin(e[3 * a + c], e[3 * b + c], 1.0, G[g]);
        } else {
            // det(E): e0 (e4 e8 - e5 e7) - e1 (e3 e8 - e5 e6) + e2 (e3 e7 - e4 e6)
            double (&t)[20] = w.t;
            for (int m = 0; m < 20; ++m) M[0][m] = 0.0;
            const int tri[3][5] = { { 0, 4, 8, 5, 7 }, { 1, 3, 8, 5, 6 }, { 2, 3, 7, 4, 6 } };
            for (int s = 0; s < 3; ++s) {
                for (int m = 0; m < 20; ++m) t[m] = 0.0;
                fpt_mul_lin_lin(e[tri[s][1]], e[tri[s][2]], 1.0, t);
                fpt_mul_lin_lin(e[tri[s][3]], e[tri[s][4]], -1.0, t);
                fpt_mul_quad_lin(t, e[tri[s][0]], s == 1 ? -1.0 : 1.0, M[0]);
            }
        }
    }
    FPT_SYNC();
    FPT_PAR_FOR(m, 0, 19) tr[m] = G[0][m] + G[4][m] + G[8][m];
    FPT_SYNC();
    FPT_PAR_FOR(g, 0, 8) {
        // C = 2 G E - tr E
        const int a = g / 3, b = g - 3 * a;
        double* row_ = M[1 + g];
        for (int m = 0; m < 20; ++m) row_[m] = 0.0;
        for (int c = 0; c < 3; ++c) fpt_mul_quad_lin(G[3 * a + c], e[3 * c + b], 2.0, row_);
        fpt_mul_quad_lin(tr, e[3 * a + b], -1.0, row_);
    }
    FPT_SYNC();
    double (&M0)[10][20] = w.M0;             // the constraints before elimination: used to polish the roots
    for (int r = 0; r < 10; ++r) {
        double nr = 0.0;
        for (int m = 0; m < 20; ++m) nr = fabs(M[r][m]) > nr ? fabs(M[r][m]) : nr;
        nr = nr > 0.0 ? 1.0 / nr : 1.0;
        for (int m = 0; m < 20; ++m) M0[r][m] = M[r][m] * nr;
    }
    // ---- 3. Gauss-Jordan on the 10 cubic monomials (on the device the 200 entries of an elimination step are
    //         spread over the lanes; the multipliers are read before anything is written)
    for (int c = 0; c < 10; ++c) {
        int p = c;
        double best = fabs(M[c][c]);
        for (int r = c + 1; r < 10; ++r) if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); p = r; }
        if (best < 1e-14) return 0;
        FPT_SYNC();
        if (p != c) { FPT_PAR_FOR(k, 0, 19) { const double t = M[c][k]; M[c][k] = M[p][k]; M[p][k] = t; } }
        FPT_SYNC();
        const double inv = 1.0 / M[c][c];
        FPT_SYNC();
        FPT_PAR_FOR(k, 0, 19) M[c][k] *= inv;
        FPT_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
        double f[4];
        for (int pass = 0; pass < 4; ++pass) {
            const int idx = (int)threadIdx.x + 64 * pass;
            f[pass] = idx < 200 ? M[idx / 20][c] : 0.0;
        }
        FPT_SYNC();
        for (int pass = 0; pass < 4; ++pass) {
            const int idx = (int)threadIdx.x + 64 * pass;
            if (idx >= 200) continue;
            const int r = idx / 20, k = idx - 20 * r;
            if (r != c && f[pass] != 0.0) M[r][k] -= f[pass] * M[c][k];
        }
        FPT_SYNC();
#else
        for (int r = 0; r < 10; ++r) {
            if (r == c) continue;
            const double f = M[r][c];
            if (f == 0.0) continue;
            for (int k = 0; k < 20; ++k) M[r][k] -= f * M[c][k];
        }
#endif
    }
    // action matrix of multiplication by x on b = [x2 xy xz y2 yz z2 x y z 1]
    double (&Ax)[100] = w.Ax;
    for (int i = 0; i < 100; ++i) Ax[i] = 0.0;
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 10; ++c) Ax[10 * r + c] = -M[r][10 + c];   // x3, x2y, x2z, xy2, xyz, xz2
    Ax[10 * 6 + 0] = 1.0; Ax[10 * 7 + 1] = 1.0; Ax[10 * 8 + 2] = 1.0; Ax[10 * 9 + 6] = 1.0;
    // ---- 4. eigenvalues of Ax: Hessenberg form by stabilised elimination, then double-shift QR in real
    //         arithmetic (Francis steps, deflation) -- no characteristic polynomial
    double (&zr)[10] = w.zr;
    double (&zi)[10] = w.zi;
    fpt_eigenvalues10(Ax, zr, zi, w);

    // ---- real roots -> eigenvectors by inverse iteration on Ax, refined eigenvalue by the eigen-equation, polished
    //      on the constraints: one root per lane on the device, in turn on the host
    double anorm = 0.0;
    for (int i = 0; i < 100; ++i) anorm = fabs(Ax[i]) > anorm ? fabs(Ax[i]) : anorm;
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
    if (threadIdx.x < 10) fpt_root_candidate(w, (int)threadIdx.x, anorm);
    __syncthreads();
#else
    for (int k = 0; k < 10; ++k) fpt_root_candidate(w, k, anorm);
#endif
    // compaction in eigenvalue order, skipping duplicates of an already accepted root
    int ns = 0;
    for (int k = 0; k < 10; ++k) {
        if (!w.root[k].valid) continue;
        const double* cand = w.root[k].cand;
        double scale = 0.0;
        for (int c = 0; c < 9; ++c) scale = fmax(scale, fabs(cand[c]));
        bool dup = false;
        for (int s2 = 0; s2 < ns; ++s2) {
            double d = 0.0;
            for (int c = 0; c < 9; ++c) d = fmax(d, fabs(cand[c] - E_out[9 * s2 + c]));
            if (d < 1e-9 * (1.0 + scale)) dup = true;
        }
        if (dup) continue;
        for (int c = 0; c < 9; ++c) E_out[9 * ns + c] = cand[c];
        ++ns;
    }
    return ns;
}

#endif
