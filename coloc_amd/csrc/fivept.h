// fivept.h -- calibrated five-point relative pose: up to 10 essential matrices from 5 correspondences.
//
// Role: hypothesis generator for the batched essential-matrix RANSAC / AC-RANSAC (clc_essential_ransac,
// clc_essential_acransac), the GPU form of RobustMatcher::filterEssential (reference include/coloc/RobustMatcher.hpp:153-186),
// which asks OpenMVG for essential::kernel::FivePointSolver inside AC-RANSAC.  OpenMVG is an empty, unpinned submodule in
// the reference snapshot, so this follows the published method (Nister 2004 / Stewenius et al. 2006), derived from scratch
// and built by polynomial arithmetic rather than by a transcribed coefficient table:
//   1. the 5 epipolar constraints q2^T E q1 = 0 leave a 4-D null space {E1..E4}: E = x E1 + y E2 + z E3 + E4;
//   2. det(E) = 0 and 2 E E^T E - trace(E E^T) E = 0 are 10 cubic polynomials in (x, y, z) over 20 monomials,
//      ordered [x3 x2y x2z xy2 xyz xz2 y3 y2z yz2 z3 | x2 xy xz y2 yz z2 x y z 1];
//   3. Gauss-Jordan on the 10 cubic monomials gives [I | B]; with basis b = [x2 xy xz y2 yz z2 x y z 1]^T the
//      action matrix of "multiply by x" has rows -B[x3], -B[x2y], -B[x2z], -B[xy2], -B[xyz], -B[xz2] and the
//      unit rows x*x = x2, x*y = xy, x*z = xz, x*1 = x;  A b = x b at every solution;
//   4. eigenvalues: Hessenberg form by stabilised elimination, balancing, then all ten at once by Ehrlich-Aberth iteration
//      on det(H - z I) evaluated with Hyman's recurrence (no characteristic-polynomial coefficients: a Faddeev-LeVerrier +
//      Durand-Kerner route lost 15 % of the true solutions to cancellation; the sequential double-shift QR iteration used
//      until round 2 had the same hit rate but is one long dependent chain -- 46 % of the GPU solve);
//   5. each real eigenvalue x gives (y, z) from six linear equations of the eigen-relation, and (x, y, z) is then polished on
//      the ten cubic constraints themselves (Gauss-Newton, 3 unknowns).
// fp64 throughout.  Inputs are NORMALISED image coordinates (K^-1 applied).
//
// This file is the sequential statement of the algorithm (it runs on the host in tests/test_fivept_host.py and
// tools/archive/fivept_host.cpp) plus the per-element math shared with the GPU form; csrc/fivept_wave.h lays the same steps out over
// the 64 lanes of one wave.
#ifndef CLC_FIVEPT_H
#define CLC_FIVEPT_H

#include <math.h>

// Fused multiply-adds throughout (the library's default is -ffp-contract=off, for the kernels whose results are compared bit for
// bit with the oracle; nothing here is: the solutions are checked by their defining properties and against the host build of
// this same file to 1e-7).  Restored at the end of csrc/fivept_wave.h / of this file.
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif

#if defined(__HIPCC__) || defined(__HIP__)
#define FPT_HD __host__ __device__ inline
#else
#define FPT_HD static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define FPT_UNROLL _Pragma("unroll")
#else
#define FPT_UNROLL
#endif

// Every array of the solver.  On the device ONE problem runs per wave and this struct lives in LDS; on the host it is a
// local variable.
struct FptWorkspace {
    double EE[4][9];                                              // null-space basis
    double G[9][20], M[10][20], M0[10][20];                       // E E^T; the constraints; the constraints with unit row maximum
    double t[20];
    double Ax[100];                                               // action matrix
    double hr[10][10];                                            // its Hessenberg form
    double zr[10], zi[10];                                        // eigenvalues
    double isub[10];                                              // Aberth: reciprocals of the subdiagonal
    double zstep[10];
    int zdone[10];
    int bal[10];                                                  // balancing exponents
    struct Root {
        double ps[6][9];                                          // device: partial Gauss-Newton sums of the six lanes of a root
        double cand[9];                                           // the essential matrix this root gives
        int valid;
    } root[10];
};

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
// ---- structured polynomial products -------------------------------------------------------------------------
// The entries of E are LINEAR in (x, y, z) (monomials 16..19), the entries of E E^T QUADRATIC (monomials 10..19):
// the only products the constraints need are linear x linear (16 terms) and quadratic x linear (40 terms), with the
// target monomial of every term known at compile time.  (The generic 20 x 20 loop with exponent look-ups that this
// replaces was half of the solver's run time.)
// c += scale * a * b, a and b linear (coefficients at 16..19), c quadratic
FPT_HD void fpt_mul_lin_lin(const double* a, const double* b, double scale, double* c)
{
FPT_UNROLL
    for (int m = 16; m < 20; ++m) {
        const double am = scale * a[m];
FPT_UNROLL
        for (int n = 16; n < 20; ++n) c[fpt_prod_index_c(m, n)] += am * b[n];
    }
}
// c += scale * a * b, a quadratic (coefficients at 10..19), b linear, c cubic
FPT_HD void fpt_mul_quad_lin(const double* a, const double* b, double scale, double* c)
{
FPT_UNROLL
    for (int m = 10; m < 20; ++m) {
        const double am = scale * a[m];
FPT_UNROLL
        for (int n = 16; n < 20; ++n) c[fpt_prod_index_c(m, n)] += am * b[n];
    }
}

// 1 / x: on the device v_rcp_f64 refined by two Newton steps (full precision for normal operands, none of the IEEE
// division sequence's scaling and fix-up -- the operands here are never subnormal or huge); a plain division on the host
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fpt_rcp(const double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// ~2^-24 relative: enough where the value only steers an iteration and does not move its fixed point
__device__ __forceinline__ double fpt_rcp_rough(const double x) { return __builtin_amdgcn_rcp(x); }
#else
FPT_HD double fpt_rcp(const double x) { return 1.0 / x; }
FPT_HD double fpt_rcp_rough(const double x) { return 1.0 / x; }
#endif
// phase time stamps for tools/archive/fivept_bench.hip (no-op in the library)
#ifndef FPT_STAMP
#define FPT_STAMP(i) ((void)0)
#endif
#ifndef FPT_SWEEP_HOOK
#define FPT_SWEEP_HOOK(it) ((void)0)
#endif


// Element loops shared by the host and the wave form: on the device the wave that owns the problem spreads them over its
// lanes, on the host they are ordinary loops.  The arithmetic per element is the same either way.
#if defined(__HIP_DEVICE_COMPILE__)
#define FPT_PAR_FOR(j, lo, hi) for (int j = (lo) + (int)threadIdx.x, once_ = 1; once_ && j <= (hi); once_ = 0)
#define FPT_SYNC() __syncthreads()
#else
#define FPT_PAR_FOR(j, lo, hi) for (int j = (lo); j <= (hi); ++j)
#define FPT_SYNC() ((void)0)
#endif
// N = p(z) / p'(z) for p(z) = det(H - z I), H upper Hessenberg (w.hr): Hyman's recurrence -- solve (H - z I) x = alpha e_1
// with x_9 = 1 from the bottom row up (each row gives x_{i-1} through the subdiagonal entry), differentiate the same
// recurrence for x', then p is proportional to alpha and p'/p = alpha'/alpha.  No polynomial coefficients are formed, so
// nothing cancels.  isub[i] = 1 / h[i][i-1] (guarded).
FPT_HD void fpt_hyman_newton(const double (&h)[10][10], const double (&isub)[10], const double zr, const double zi, double* nr, double* ni)
{
    // column-oriented: as soon as x_j is known its contribution goes into the running sums of ALL rows above, so the only
    // dependent chain is x_j -> (diagonal term of row j) -> x_{j-1}; the forty running sums are independent of one another
    // (a row-by-row dot product is one long chain of dependent fp64 FMAs: measured 2x slower per sweep on the GPU)
    double sr[10], si[10], tr[10], ti[10];
FPT_UNROLL
    for (int i = 0; i < 10; ++i) { sr[i] = 0.0; si[i] = 0.0; tr[i] = 0.0; ti[i] = 0.0; }
    double xr = 1.0, xi = 0.0, dr = 0.0, di = 0.0;
FPT_UNROLL
    for (int j = 9; j >= 0; --j) {
        // rows above the diagonal: real coefficient
FPT_UNROLL
        for (int i = 0; i < j; ++i) { const double hij = h[i][j]; sr[i] += hij * xr; si[i] += hij * xi; tr[i] += hij * dr; ti[i] += hij * di; }
        // row j: (h_jj - z) x_j, and -x_j for the derivative
        const double hr_ = h[j][j] - zr;
        const double ar = sr[j] + (hr_ * xr + zi * xi), ai = si[j] + (hr_ * xi - zi * xr);
        const double br = tr[j] + (hr_ * dr + zi * di) - xr, bi = ti[j] + (hr_ * di - zi * dr) - xi;
        if (j > 0) { const double m = -isub[j]; xr = ar * m; xi = ai * m; dr = br * m; di = bi * m; }
        else { sr[0] = ar; si[0] = ai; tr[0] = br; ti[0] = bi; }
    }
    double ar = sr[0], ai = si[0], br = tr[0], bi = ti[0];
    // alpha / alpha' with the denominator scaled to unit size first
    const double sc = fmax(fabs(br), fabs(bi));
    if (!(sc > 0.0) || !(sc < 1e300)) { *nr = 0.0; *ni = 0.0; return; }
    const double inv = fpt_rcp(sc);
    br *= inv; bi *= inv; ar *= inv; ai *= inv;
    const double d = fpt_rcp(br * br + bi * bi);
    *nr = (ar * br + ai * bi) * d;
    *ni = (ai * br - ar * bi) * d;
}


// Balancing: B = D^-1 H D with D = diag(2^e_i) (exact, keeps the Hessenberg form and the eigenvalues).  The elimination that
// produced H is not orthogonal and leaves entries of 1e7..1e10 next to eigenvalues of order 1..100; the rounding noise of
// every later evaluation scales with the norm.  Three simultaneous sweeps: each row/column pair picks the power of two that
// brings its off-diagonal column and row sums together (element (i,j) is scaled by 2^(e_j - e_i)).  Measured on 5000
// synthetic scenes: 14.1 Aberth sweeps on average with it, 17.5 without.
FPT_HD void fpt_balance10(FptWorkspace& w)
{
    const int n = 10;
    double (&a)[10][10] = w.hr;
    FPT_SYNC();
    for (int sweep = 0; sweep < 3; ++sweep) {
        FPT_PAR_FOR(i, 0, n - 1) {
            double cn = 0.0, rn = 0.0;
            for (int j = 0; j < n; ++j) if (j != i) { cn += fabs(a[j][i]); rn += fabs(a[i][j]); }
            int e = 0;
            if (cn > 0.0 && rn > 0.0) {
                int ec, er;
                (void)frexp(cn, &ec);
                (void)frexp(rn, &er);
                e = (er - ec) / 2;                               // d_i = 2^e: column i times d_i, row i divided by d_i
            }
            w.bal[i] = e;
        }
        FPT_SYNC();
        FPT_PAR_FOR(idx, 0, 99) { const int i = idx / 10, j = idx - 10 * i; a[i][j] = ldexp(a[i][j], w.bal[j] - w.bal[i]); }
#if defined(__HIP_DEVICE_COMPILE__)
        FPT_PAR_FOR(idx, 64, 99) { const int i = idx / 10, j = idx - 10 * i; a[i][j] = ldexp(a[i][j], w.bal[j] - w.bal[i]); }
#endif
        FPT_SYNC();
    }
}
#if defined(__HIP_DEVICE_COMPILE__)
// the partner lane's value inside an (even, odd) lane pair: DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ double fpt_pair_swap(const double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// fpt_hyman_newton on TWO lanes per root: the even lane carries the real parts of x, x' and of the forty running sums, the odd
// lane the imaginary parts.  Above the diagonal the coefficients are real, so each lane updates its own half with the same
// instructions (2 FMAs per entry instead of 4); only the diagonal term (h_jj - z) x_j mixes the halves -- one DPP swap of x and
// x' per column.  Same operations on the same operands as the one-lane form, both lanes end with the full N.
// (210 instead of 356 fp64 instructions per evaluation: the iteration is issue-bound, one wave per SIMD.)
__device__ __forceinline__ void fpt_hyman_newton_pair(const double (&h)[10][10], const double (&isub)[10], const double zr, const double zi,
                                                      const int comp, double* nr, double* ni)
{
    double s[10], t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { s[i] = 0.0; t[i] = 0.0; }
    const double szi = comp ? -zi : zi;
    double x = comp ? 0.0 : 1.0, d = 0.0, alpha = 0.0, dalpha = 0.0;
#pragma unroll
    for (int j = 9; j >= 0; --j) {
#pragma unroll
        for (int i = 0; i < j; ++i) { const double hij = h[i][j]; s[i] += hij * x; t[i] += hij * d; }
        const double px = fpt_pair_swap(x), pd = fpt_pair_swap(d);
        const double hr_ = h[j][j] - zr;
        const double a = s[j] + (hr_ * x + szi * px);
        const double b = t[j] + (hr_ * d + szi * pd) - x;
        if (j > 0) { const double m = -isub[j]; x = a * m; d = b * m; }
        else { alpha = a; dalpha = b; }
    }
    const double palpha = fpt_pair_swap(alpha), pdalpha = fpt_pair_swap(dalpha);
    double ar = comp ? palpha : alpha, ai = comp ? alpha : palpha, br = comp ? pdalpha : dalpha, bi = comp ? dalpha : pdalpha;
    const double sc = fmax(fabs(br), fabs(bi));
    if (!(sc > 0.0) || !(sc < 1e300)) { *nr = 0.0; *ni = 0.0; return; }
    const double inv = fpt_rcp(sc);
    br *= inv; bi *= inv; ar *= inv; ai *= inv;
    const double dd = fpt_rcp(br * br + bi * bi);
    *nr = (ar * br + ai * bi) * dd;
    *ni = (ai * br - ar * bi) * dd;
}
#endif

// All ten eigenvalues of the Hessenberg matrix in w.hr at once: Ehrlich-Aberth iteration
//     z_k <- z_k - N_k / (1 - N_k sum_{j != k} 1 / (z_k - z_j)),   N_k = p(z_k) / p'(z_k)  (fpt_hyman_newton),
// every root updated from the previous sweep's values (on the device: one root per lane, ten lanes, state in registers, the
// other roots over v_readlane; on the host the same sweeps in a loop).  Cubic convergence for simple roots once near.
// Start: the DIAGONAL of the balanced Hessenberg matrix, each entry pushed off the real axis by 3 % of its size (alternating
// sign, slightly different per root).  The action matrix's eigenvalues spread over orders of magnitude; points on one circle
// spend ~10 ln(ratio) sweeps creeping inwards (53 sweeps on average, measured on 5000 synthetic scenes), a geometric ladder
// of radii around |det|^(1/10) 14, the diagonal 10.4 -- and, what matters for a kernel that waits for its slowest wave, the
// tail goes: 99 % of the problems are done after 19 sweeps instead of 32.  A root stops when its step is below 1e-12 of its
// size, or small and no longer shrinking (evaluation noise of an ill-conditioned eigenvalue); everything stops after 16
// sweeps (hit rate 0.994; 18 sweeps give 0.9948, 14 give 0.990; the double-shift QR iteration used until round 2 had 0.9942).  Roots are polished on the
// constraints afterwards.  (The sequential QR took 46 % of the solver's time on the GPU: ~150 dependent Householder steps
// of divisions and square roots, each with LDS round trips.)
#ifndef FPT_CPLX_STAG
#define FPT_CPLX_STAG 1e-4
#endif
#ifndef FPT_SWEEPS
#define FPT_SWEEPS 16
#endif
constexpr int kFptAberthSweeps = FPT_SWEEPS;
// Only the REAL eigenvalues are used.  A root that is plainly complex and already creeping (step below FPT_CPLX_SETTLE of its
// size) no longer holds the iteration up: where exactly it ends does not change the real ones' limits.
#ifndef FPT_CPLX_SETTLE
#define FPT_CPLX_SETTLE 1e-3
#endif
FPT_HD bool fpt_aberth_settled_complex(const double zr, const double zi, const double step)
{
    const double zm = fabs(zr) + fabs(zi);
    return fabs(zi) > 1e-2 * zm && step <= FPT_CPLX_SETTLE * zm;
}
FPT_HD void fpt_aberth10(FptWorkspace& w, double* wr, double* wi, const double anorm)
{
    const double (&h)[10][10] = w.hr;
    double (&isub)[10] = w.isub;
    const double tiny = 1e-14 * anorm;
    FPT_SYNC();
    FPT_PAR_FOR(k, 0, 9) {
        const double s = h[k > 0 ? k : 1][k > 0 ? k - 1 : 0];
        isub[k] = 1.0 / (fabs(s) > tiny ? s : (s < 0.0 ? -tiny : tiny));
        const double d = h[k][k];
        wr[k] = d;
        wi[k] = (k & 1 ? -0.03 : 0.03) * (1.0 + 0.1 * (double)k) * (fabs(d) + 1e-12 * anorm);
        w.zdone[k] = 0;
        w.zstep[k] = 1e300;
    }
    FPT_SYNC();
#if defined(__HIP_DEVICE_COMPILE__)
    // the upper triangle + subdiagonal reciprocals in registers for the whole iteration (left in LDS they are re-read after
    // every barrier: ~100 ds_read per sweep)
    double hreg[10][10], isubreg[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        isubreg[i] = isub[i];
#pragma unroll
        for (int j = 0; j < 10; ++j) hreg[i][j] = j >= i ? h[i][j] : 0.0;
    }
    // made opaque AFTER all the loads are issued: otherwise the compiler re-reads LDS inside the iteration instead of keeping
    // the 64 values in registers
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        asm volatile("" : "+v"(isubreg[i]));
#pragma unroll
        for (int j = i; j < 10; ++j) asm volatile("" : "+v"(hreg[i][j]));
    }
#define FPT_H hreg
#define FPT_ISUB isubreg
#else
#define FPT_H h
#define FPT_ISUB isub
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    {
        // one root per lane PAIR (real / imaginary halves of the Hyman evaluation, everything else computed identically by both),
        // state in registers; the other roots' positions come over v_readlane (lane index static) and the stop test is a ballot:
        // no LDS traffic and no barrier inside the iteration
        const int k = (int)threadIdx.x < 20 ? (int)threadIdx.x >> 1 : 9;
        const int comp = (int)threadIdx.x & 1;
        const bool mine = (int)threadIdx.x < 20;
        double zr = wr[k], zi = wi[k], step = 1e300;
        bool done = !mine;
        for (int it = 0; it < kFptAberthSweeps; ++it) {
            double or_[10], oi_[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                or_[j] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(zr), 2 * j), __builtin_amdgcn_readlane(__double2loint(zr), 2 * j));
                oi_[j] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(zi), 2 * j), __builtin_amdgcn_readlane(__double2loint(zi), 2 * j));
            }
            if (!done) {
                double nr, ni;
                fpt_hyman_newton_pair(FPT_H, FPT_ISUB, zr, zi, comp, &nr, &ni);
                double sr = 0.0, si = 0.0;
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const double dr = zr - or_[j], di = zi - oi_[j];
                    const double d2 = dr * dr + di * di;
                    const double q = (j != k && d2 > 0.0) ? fpt_rcp_rough(d2 > 0.0 ? d2 : 1.0) : 0.0;
                    sr += dr * q; si -= di * q;
                }
                const double er = 1.0 - (nr * sr - ni * si), ei = -(nr * si + ni * sr);
                const double e2 = er * er + ei * ei;
                double dlr = nr, dli = ni;
                if (e2 > 0.0) { const double q = fpt_rcp(e2); dlr = (nr * er + ni * ei) * q; dli = (ni * er - nr * ei) * q; }
                if (!(dlr == dlr) || !(dli == dli)) { dlr = 0.0; dli = 0.0; done = true; }
                zr -= dlr; zi -= dli;
                const double dm = fabs(dlr) + fabs(dli), zm = fabs(zr) + fabs(zi);
                if (dm <= 1e-12 * zm + 1e-300) done = true;
                else if (dm <= (fabs(zi) > 1e-4 * zm ? FPT_CPLX_STAG : 1e-7) * zm && dm >= 0.5 * step) done = true;
                step = dm;
            }
            if (__ballot(!done && !fpt_aberth_settled_complex(zr, zi, step)) == 0ull) break;
            FPT_STAMP(10 + (it >= 5 ? 5 : it));
        }
        if (mine && comp == 0) { wr[k] = zr; wi[k] = zi; }
    }
#else
    for (int it = 0; it < kFptAberthSweeps; ++it) {
        double nzr[10], nzi[10], nstep[10];
        int ndone[10];
#define FPT_SLOT(k) (k)
        FPT_PAR_FOR(k, 0, 9) {
            const double zr = wr[k], zi = wi[k];
            double new_r = zr, new_i = zi, step = w.zstep[k];
            int done = w.zdone[k];
            if (!done) {
                double nr, ni;
                fpt_hyman_newton(FPT_H, FPT_ISUB, zr, zi, &nr, &ni);
                double sr = 0.0, si = 0.0;
FPT_UNROLL
                for (int j = 0; j < 10; ++j) {                   // branch-free: the nine reciprocals overlap
                    const double dr = zr - wr[j], di = zi - wi[j];
                    const double d2 = dr * dr + di * di;
                    const double q = (j != k && d2 > 0.0) ? fpt_rcp_rough(d2 > 0.0 ? d2 : 1.0) : 0.0;
                    sr += dr * q; si -= di * q;
                }
                // delta = N / (1 - N S)
                const double er = 1.0 - (nr * sr - ni * si), ei = -(nr * si + ni * sr);
                const double e2 = er * er + ei * ei;
                double dlr = nr, dli = ni;
                if (e2 > 0.0) { const double q = fpt_rcp(e2); dlr = (nr * er + ni * ei) * q; dli = (ni * er - nr * ei) * q; }
                if (!(dlr == dlr) || !(dli == dli)) { dlr = 0.0; dli = 0.0; done = 1; }
                new_r = zr - dlr; new_i = zi - dli;
                const double dm = fabs(dlr) + fabs(dli), zm = fabs(new_r) + fabs(new_i);
                if (dm <= 1e-12 * zm + 1e-300) done = 1;
                else if (dm <= (fabs(new_i) > 1e-4 * zm ? FPT_CPLX_STAG : 1e-7) * zm && dm >= 0.5 * step) done = 1;
                step = dm;
            }
            nzr[FPT_SLOT(k)] = new_r; nzi[FPT_SLOT(k)] = new_i; nstep[FPT_SLOT(k)] = step; ndone[FPT_SLOT(k)] = done;
        }
        FPT_SYNC();
        FPT_PAR_FOR(k, 0, 9) { wr[k] = nzr[FPT_SLOT(k)]; wi[k] = nzi[FPT_SLOT(k)]; w.zstep[k] = nstep[FPT_SLOT(k)]; w.zdone[k] = ndone[FPT_SLOT(k)]; }
#undef FPT_SLOT
        FPT_SYNC();
        int all = 1;
        for (int k = 0; k < 10; ++k) all &= (w.zdone[k] || fpt_aberth_settled_complex(wr[k], wi[k], w.zstep[k]));
        FPT_SWEEP_HOOK(it);
        if (all) break;
        FPT_STAMP(10 + (it >= 5 ? 5 : it));
    }
#endif
#undef FPT_H
#undef FPT_ISUB
    FPT_SYNC();
}


// Hessenberg form of w.hr by elimination with row/column interchanges (similarity transforms), sequential statement
// (fivept_wave.h does a whole column step per pass)
FPT_HD void fpt_hessenberg10_seq(FptWorkspace& w)
{
    const int n = 10;
    double (&a)[10][10] = w.hr;
    for (int m = 1; m < n - 1; ++m) {
        int p = m;
        double x = 0.0;
        for (int j = m; j < n; ++j) if (fabs(a[j][m - 1]) > fabs(x)) { x = a[j][m - 1]; p = j; }
        if (p != m) {
            for (int j = m - 1; j < n; ++j) { const double t = a[p][j]; a[p][j] = a[m][j]; a[m][j] = t; }
            for (int j = 0; j < n; ++j) { const double t = a[j][p]; a[j][p] = a[j][m]; a[j][m] = t; }
        }
        if (x != 0.0) {
            double ys[10];
            for (int i = m + 1; i < n; ++i) {
                ys[i] = a[i][m - 1] / x;
                for (int j = m; j < n; ++j) a[i][j] -= ys[i] * a[m][j];
            }
            for (int j = 0; j < n; ++j) {
                double acc = a[j][m];
                for (int i = m + 1; i < n; ++i) acc += ys[i] * a[j][i];
                a[j][m] = acc;
                if (j > m) a[j][m - 1] = 0.0;
            }
        }
    }
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


// ---- roots ---------------------------------------------------------------------------------------------------------------
FPT_HD bool fpt_root_is_real(const FptWorkspace& w, const int k) { return !(fabs(w.zi[k]) > 1e-6 * (1.0 + fabs(w.zr[k]))); }

// Real eigenvalue k of the action matrix -> starting point (x, y, z) of the polish; false if the eigenvalue is complex or
// the linear system is singular.
FPT_HD bool fpt_root_start(const FptWorkspace& w, const int k, double* x_out, double* y_out, double* z_out)
{
    if (!fpt_root_is_real(w, k)) return false;
    const double (&Ax)[100] = w.Ax;
    const double (&zr)[10] = w.zr;
    {
        // (y, z) straight from the eigenvalue: at a solution the basis vector is b = [x2 xy xz y2 yz z2 x y z 1] with x = lambda,
        // and rows 0..5 of (Ax - lambda I) b = 0 are six equations LINEAR in u = (y, z, y2, yz, z2) -- the rows 6..9 are the
        // identities x*x = x2, x*y = xy, x*z = xz, x*1 = x that fixed b[0..2] and b[6].  Gaussian elimination with row pivoting
        // on the 6 x 6 augmented system (five of the six rows are used, the pivoting picks them), all indices static so the
        // 36 values live in registers.  The polish below then works on the constraints themselves.  (Two steps of inverse
        // iteration on Ax -- 10 x 10 LU with dynamic pivoting -- did the same job until round 2 at 13 % of the solver's time.)
        const double lam = zr[k];
        double C[6][6];
FPT_UNROLL
        for (int r = 0; r < 6; ++r) {
            const double* a = Ax + 10 * r;
            C[r][0] = a[1] * lam + a[7];                // y
            C[r][1] = a[2] * lam + a[8];                // z
            C[r][2] = a[3];                             // y2
            C[r][3] = a[4];                             // yz
            C[r][4] = a[5];                             // z2
            C[r][5] = -((a[0] * lam + a[6]) * lam + a[9]);   // right-hand side: -(a0 x2 + a6 x + a9)
        }
        C[0][5] += lam * lam * lam;                     // - lambda b[r] on the left: b[0] = x2
        C[1][0] -= lam * lam;                           //                             b[1] = x y
        C[2][1] -= lam * lam;                           //                             b[2] = x z
        C[3][2] -= lam; C[4][3] -= lam; C[5][4] -= lam; //                             b[3..5] = y2, yz, z2
FPT_UNROLL
        for (int c = 0; c < 5; ++c) {
            int p = c;
            double best = fabs(C[c][c]);
FPT_UNROLL
            for (int r = c + 1; r < 6; ++r) if (fabs(C[r][c]) > best) { best = fabs(C[r][c]); p = r; }
            if (!(best > 0.0)) return false;
FPT_UNROLL
            for (int r = c + 1; r < 6; ++r) {
                const bool sw = p == r;
FPT_UNROLL
                for (int j = c; j < 6; ++j) { const double t = C[c][j]; C[c][j] = sw ? C[r][j] : t; C[r][j] = sw ? t : C[r][j]; }
            }
            const double inv = 1.0 / C[c][c];
FPT_UNROLL
            for (int r = c + 1; r < 6; ++r) {
                const double f = C[r][c] * inv;
FPT_UNROLL
                for (int j = c + 1; j < 6; ++j) C[r][j] -= f * C[c][j];
            }
        }
        double u[5];
FPT_UNROLL
        for (int r = 4; r >= 0; --r) {
            double acc = C[r][5];
FPT_UNROLL
            for (int j = r + 1; j < 5; ++j) acc -= C[r][j] * u[j];
            u[r] = acc / C[r][r];
        }
        if (!(u[0] == u[0]) || !(u[1] == u[1])) return false;
        *x_out = lam; *y_out = u[0]; *z_out = u[1];
    }
    return true;
}

// the 20 monomials at (x, y, z), compile-time exponents
FPT_HD void fpt_monomials(const double x, const double y, const double z, double (&mono)[20])
{
    const double px[4] = { 1.0, x, x * x, x * x * x }, py[4] = { 1.0, y, y * y, y * y * y }, pz[4] = { 1.0, z, z * z, z * z * z };
FPT_UNROLL
    for (int m = 0; m < 20; ++m) mono[m] = px[kFptExp[m][0]] * py[kFptExp[m][1]] * pz[kFptExp[m][2]];
}

// One constraint row of a Gauss-Newton step: residual rr and gradient (jx, jy, jz) of sum_m c_m mono_m, added into
// sums = [JtJ00 JtJ01 JtJ02 JtJ11 JtJ12 JtJ22 | Jtr0 Jtr1 Jtr2].  The derivative of a monomial is (exponent) x (another
// monomial of the same set), so the 20 monomial values are all that is kept (keeping the 60 derivative values as well
// pushed the kernel to 512 VGPRs + scratch).
FPT_HD void fpt_polish_row(const double* coef, const double (&mono)[20], double (&sums)[9])
{
    double rr = 0.0, jx = 0.0, jy = 0.0, jz = 0.0;
FPT_UNROLL
    for (int m = 0; m < 20; ++m) {
        const int ei = kFptExp[m][0], ej = kFptExp[m][1], ek = kFptExp[m][2];
        const double cm = coef[m];
        rr += cm * mono[m];
        if (ei) jx += (ei * cm) * mono[fpt_mono_index_c(ei ? ei - 1 : 0, ej, ek)];
        if (ej) jy += (ej * cm) * mono[fpt_mono_index_c(ei, ej ? ej - 1 : 0, ek)];
        if (ek) jz += (ek * cm) * mono[fpt_mono_index_c(ei, ej, ek ? ek - 1 : 0)];
    }
    sums[0] += jx * jx; sums[1] += jx * jy; sums[2] += jx * jz; sums[3] += jy * jy; sums[4] += jy * jz; sums[5] += jz * jz;
    sums[6] += jx * rr; sums[7] += jy * rr; sums[8] += jz * rr;
}

// Solve the 3 x 3 normal equations and step.  0 = stepped, go on; 1 = stepped and converged; -1 = singular / not finite (no step).
FPT_HD int fpt_polish_update(const double (&s)[9], double* x, double* y, double* z)
{
    const double a00 = s[0], a01 = s[1], a02 = s[2], a11 = s[3], a12 = s[4], a22 = s[5], b0 = s[6], b1 = s[7], b2 = s[8];
    const double det = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
    if (!(fabs(det) > 1e-300)) return -1;
    const double idet = fpt_rcp(det);
    const double dx = (b0 * (a11 * a22 - a12 * a12) - a01 * (b1 * a22 - a12 * b2) + a02 * (b1 * a12 - a11 * b2)) * idet;
    const double dy = (a00 * (b1 * a22 - a12 * b2) - b0 * (a01 * a22 - a12 * a02) + a02 * (a01 * b2 - b1 * a02)) * idet;
    const double dz = (a00 * (a11 * b2 - b1 * a12) - a01 * (a01 * b2 - b1 * a02) + b0 * (a01 * a12 - a11 * a02)) * idet;
    if (!(dx == dx) || !(dy == dy) || !(dz == dz)) return -1;
    *x -= dx; *y -= dy; *z -= dz;
    return fabs(dx) + fabs(dy) + fabs(dz) < 1e-15 * (1.0 + fabs(*x) + fabs(*y) + fabs(*z)) ? 1 : 0;
}

// (x, y, z) -> candidate essential matrix w.root[k].cand; valid only if it satisfies the cubic constraints: an eigenvalue
// that was not a true real root yields a matrix that is not essential
FPT_HD void fpt_root_finish(FptWorkspace& w, const int k, const double x, const double y, const double z)
{
    FptWorkspace::Root& rw = w.root[k];
    bool finite = true;
    for (int c = 0; c < 9; ++c) {
        const double ev = x * w.EE[0][c] + y * w.EE[1][c] + z * w.EE[2][c] + w.EE[3][c];
        rw.cand[c] = ev;
        finite = finite && (ev == ev) && fabs(ev) < 1e300;
    }
    rw.valid = (finite && fpt_constraint_residual(rw.cand) < 1e-9) ? 1 : 0;
}

// compaction in eigenvalue order, skipping duplicates of an already accepted root
FPT_HD int fpt_compact(const FptWorkspace& w, double* E_out)
{
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

// ---- the sequential statement ---------------------------------------------------------------------------------------------
// q1, q2: 5 x 2 normalised coordinates in view 1 / view 2 (q2^T E q1 = 0).  E_out: up to 10 x 9 (row-major 3x3).
#if !defined(__HIP_DEVICE_COMPILE__)
static inline int fivept_solve(const double q1[5][2], const double q2[5][2], double* E_out, FptWorkspace& w)
{
    // ---- 1. null space of the 5 x 9 constraint matrix by reduced row echelon form
    double A[5][9];
    for (int i = 0; i < 5; ++i) {
        const double a[3] = { q2[i][0], q2[i][1], 1.0 }, b[3] = { q1[i][0], q1[i][1], 1.0 };
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[i][3 * r + c] = a[r] * b[c];
    }
    int piv[5], is_piv[9];
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
    // ---- 2. the ten cubic constraints.  e[k] = linear polynomial x EE0[k] + y EE1[k] + z EE2[k] + EE3[k]
    double e[9][20];
    for (int k = 0; k < 9; ++k) {
        for (int m = 0; m < 20; ++m) e[k][m] = 0.0;
        e[k][16] = EE[0][k]; e[k][17] = EE[1][k]; e[k][18] = EE[2][k]; e[k][19] = EE[3][k];
    }
    double (&M)[10][20] = w.M;
    double (&G)[9][20] = w.G;
    for (int g = 0; g < 9; ++g) {                                 // G = E E^T (quadratic)
        const int a = g / 3, b = g - 3 * a;
        for (int m = 0; m < 20; ++m) G[g][m] = 0.0;
        for (int c = 0; c < 3; ++c) fpt_mul_lin_lin(e[3 * a + c], e[3 * b + c], 1.0, G[g]);
    }
    {                                                             // det(E): e0 (e4 e8 - e5 e7) - e1 (e3 e8 - e5 e6) + e2 (e3 e7 - e4 e6)
        for (int m = 0; m < 20; ++m) M[0][m] = 0.0;
        const int tri[3][5] = { { 0, 4, 8, 5, 7 }, { 1, 3, 8, 5, 6 }, { 2, 3, 7, 4, 6 } };
        for (int s = 0; s < 3; ++s) {
            double t[20];
            for (int m = 0; m < 20; ++m) t[m] = 0.0;
            fpt_mul_lin_lin(e[tri[s][1]], e[tri[s][2]], 1.0, t);
            fpt_mul_lin_lin(e[tri[s][3]], e[tri[s][4]], -1.0, t);
            fpt_mul_quad_lin(t, e[tri[s][0]], s == 1 ? -1.0 : 1.0, M[0]);
        }
    }
    double tr[20];
    for (int m = 0; m < 20; ++m) tr[m] = G[0][m] + G[4][m] + G[8][m];
    for (int g = 0; g < 9; ++g) {                                 // C = 2 G E - tr E
        const int a = g / 3, b = g - 3 * a;
        double* row_ = M[1 + g];
        for (int m = 0; m < 20; ++m) row_[m] = 0.0;
        for (int c = 0; c < 3; ++c) fpt_mul_quad_lin(G[3 * a + c], e[3 * c + b], 2.0, row_);
        fpt_mul_quad_lin(tr, e[3 * a + b], -1.0, row_);
    }
    double (&M0)[10][20] = w.M0;             // the constraints before elimination, unit row maximum: used to polish the roots
    for (int r = 0; r < 10; ++r) {
        double nr = 0.0;
        for (int m = 0; m < 20; ++m) nr = fabs(M[r][m]) > nr ? fabs(M[r][m]) : nr;
        nr = nr > 0.0 ? 1.0 / nr : 1.0;
        for (int m = 0; m < 20; ++m) M0[r][m] = M[r][m] * nr;
    }
    // ---- 3. Gauss-Jordan on the 10 cubic monomials (the multipliers of a step are read before anything is written)
    for (int c = 0; c < 10; ++c) {
        int p = c;
        double best = fabs(M[c][c]);
        for (int r = c + 1; r < 10; ++r) if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); p = r; }
        if (best < 1e-14) return 0;
        if (p != c) for (int k = 0; k < 20; ++k) { const double t = M[c][k]; M[c][k] = M[p][k]; M[p][k] = t; }
        const double inv = 1.0 / M[c][c];
        for (int k = 0; k < 20; ++k) M[c][k] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == c) continue;
            const double f = M[r][c];
            for (int k = 0; k < 20; ++k) M[r][k] -= f * M[c][k];
        }
    }
    // action matrix of multiplication by x on b = [x2 xy xz y2 yz z2 x y z 1]
    double (&Ax)[100] = w.Ax;
    for (int i = 0; i < 100; ++i) Ax[i] = 0.0;
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 10; ++c) Ax[10 * r + c] = -M[r][10 + c];   // x3, x2y, x2z, xy2, xyz, xz2
    Ax[10 * 6 + 0] = 1.0; Ax[10 * 7 + 1] = 1.0; Ax[10 * 8 + 2] = 1.0; Ax[10 * 9 + 6] = 1.0;
    // ---- 4. eigenvalues
    for (int i = 0; i < 100; ++i) w.hr[i / 10][i % 10] = Ax[i];
    fpt_hessenberg10_seq(w);
    fpt_balance10(w);
    double anorm = 0.0;
    for (int r = 0; r < 10; ++r) for (int c = (r > 0 ? r - 1 : 0); c < 10; ++c) anorm += fabs(w.hr[r][c]);
    if (!(anorm > 0.0)) return 0;
    fpt_aberth10(w, w.zr, w.zi, anorm);
    // ---- 5. real roots -> (y, z) -> polish on the constraints
    for (int k = 0; k < 10; ++k) {
        w.root[k].valid = 0;
        double x, y, z;
        if (!fpt_root_start(w, k, &x, &y, &z)) continue;
        for (int it = 0; it < 4; ++it) {
            double mono[20], s[9] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
            fpt_monomials(x, y, z, mono);
            for (int r = 0; r < 10; ++r) fpt_polish_row(M0[r], mono, s);
            if (fpt_polish_update(s, &x, &y, &z) != 0) break;
        }
        fpt_root_finish(w, k, x, y, z);
    }
    return fpt_compact(w, E_out);
}
#endif

#if defined(__clang__) && !defined(CLC_FIVEPT_WAVE_H)
#pragma clang fp contract(off)
#endif

#endif
