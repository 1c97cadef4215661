// fivept_wave.h -- the five-point solver of fivept.h laid out for ONE 64-lane wave per problem (device only).
//
// Same algorithm, same arithmetic per element as the sequential statement in fivept.h (which stays the readable
// reference and runs on the host in tests/); what changes is where the numbers live and who touches them:
//   1. null space of the 5 x 9 constraint matrix      lane c = column c, the five rows in registers; the pivot column is
//                                                      broadcast with v_readlane (lane index uniform), row swaps are
//                                                      uniform branches -- no LDS, no barrier
//   2. the ten cubic constraints                       one polynomial per lane, accumulated in REGISTERS (target monomials
//                                                      are compile-time constants) and stored once; the old form accumulated
//                                                      in LDS, a read-modify-write chain per term
//   3. Gauss-Jordan on the 10 x 20 coefficient matrix  lane k = column k, the ten rows in registers; per step 20 v_readlane
//                                                      for the pivot column, one reciprocal, nine FMAs
//   4. Hessenberg form of the 10 x 10 action matrix    in LDS, but a whole elimination step at once: all row operations of a
//                                                      column in one pass (one element per lane), then all column updates
//                                                      (one row per lane): 4 barriers per column instead of 2 per ROW
//   5. balancing + Ehrlich-Aberth eigenvalues          fivept.h (one root per lane, registers + v_readlane)
//   6. roots -> (y, z) -> polish on the constraints    six lanes per root: each takes (at most) two of the ten constraint rows
//                                                      of a Gauss-Newton step, partial sums meet in LDS
// Why: one problem per wave is latency-bound -- a solve is a chain of ~10^4 dependent steps and every LDS round trip or
// barrier in it costs 100+ cycles.  Phase times for one problem before -> after (cycles, MI355X): null space 17.5k -> 4.5k,
// polynomials 13.3k -> 5.4k, Gauss-Jordan 27.5k -> 7.4k, Hessenberg 26.3k -> 8.9k, roots 175k -> 16.6k
// (profiles/r02_fivept_phases.txt); 256 samples: 257.7 us -> 44.5 us.
#ifndef CLC_FIVEPT_WAVE_H
#define CLC_FIVEPT_WAVE_H

#include "fivept.h"

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

namespace fpw {

__device__ __forceinline__ double readlane(const double v, const int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// ---- 1. null space: returns false for a rank-deficient sample.  On return lane c < 9 holds ee[nb] = EE[nb][c].
__device__ __forceinline__ bool null_space(const double q1[5][2], const double q2[5][2], double (&ee)[4])
{
    const int lane = (int)threadIdx.x;
    const int cc = lane < 9 ? lane : 8, rr = cc / 3, c3 = cc - 3 * rr;
    double a[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double u = rr == 0 ? q2[i][0] : (rr == 1 ? q2[i][1] : 1.0);
        const double v = c3 == 0 ? q1[i][0] : (c3 == 1 ? q1[i][1] : 1.0);
        a[i] = u * v;
    }
    int col = 0, pivc[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        int p = r;
        for (;;) {                                               // uniform: every lane sees the same broadcast values
            if (col >= 9) return false;
            double best = -1.0;
            p = r;
#pragma unroll
            for (int i = r; i < 5; ++i) { const double v = fabs(readlane(a[i], col)); if (v > best) { best = v; p = i; } }
            if (!(best < 1e-12)) break;
            ++col;
        }
#pragma unroll
        for (int i = r + 1; i < 5; ++i) if (p == i) { const double t = a[r]; a[r] = a[i]; a[i] = t; }
        const double inv = fpt_rcp(readlane(a[r], col));
        a[r] *= inv;
#pragma unroll
        for (int i = 0; i < 5; ++i) if (i != r) { const double f = readlane(a[i], col); a[i] -= f * a[r]; }
        pivc[r] = col;
        ++col;
    }
    unsigned free_cols = 0x1ffu;
#pragma unroll
    for (int r = 0; r < 5; ++r) free_cols &= ~(1u << pivc[r]);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        const int f = __builtin_ctz(free_cols);                  // free columns in ascending order
        free_cols &= free_cols - 1u;
        double e = lane == f ? 1.0 : 0.0;
#pragma unroll
        for (int r = 0; r < 5; ++r) { const double v = readlane(a[r], f); if (lane == pivc[r]) e = -v; }
        ee[nb] = e;
    }
    return true;
}

// ---- 3. Gauss-Jordan: lane k < 20 holds column k of M (ten rows).  false = a pivot below 1e-14.
__device__ __forceinline__ bool gauss_jordan(double (&m)[10])
{
#pragma unroll
    for (int c = 0; c < 10; ++c) {
        double v[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = readlane(m[i], c);
        int p = c;
        double best = fabs(v[c]);
#pragma unroll
        for (int i = c + 1; i < 10; ++i) if (fabs(v[i]) > best) { best = fabs(v[i]); p = i; }
        if (best < 1e-14) return false;
#pragma unroll
        for (int i = c + 1; i < 10; ++i)
            if (p == i) { double t = m[c]; m[c] = m[i]; m[i] = t; t = v[c]; v[c] = v[i]; v[i] = t; }
        const double inv = fpt_rcp(v[c]);
        m[c] *= inv;
#pragma unroll
        for (int i = 0; i < 10; ++i) if (i != c) m[i] -= v[i] * m[c];
    }
    return true;
}

// ---- 4. Hessenberg form of w.hr by stabilised elimination, one whole column step at a time
__device__ __forceinline__ void hessenberg(FptWorkspace& w)
{
    const int lane = (int)threadIdx.x;
    double (&a)[10][10] = w.hr;
    double (&ys)[20] = w.t;
    for (int m = 1; m < 9; ++m) {
        int p = m;
        double x = 0.0;
        for (int j = m; j < 10; ++j) { const double v = a[j][m - 1]; if (fabs(v) > fabs(x)) { x = v; p = j; } }
        __syncthreads();
        if (p != m) {
            if (lane >= m - 1 && lane < 10) { const double t = a[p][lane]; a[p][lane] = a[m][lane]; a[m][lane] = t; }
            __syncthreads();
            if (lane < 10) { const double t = a[lane][p]; a[lane][p] = a[lane][m]; a[lane][m] = t; }
            __syncthreads();
        }
        if (x != 0.0) {
            // rows m+1..9, columns m..9: a[i][j] -= (a[i][m-1] / x) a[m][j]; row m and column m-1 are only read here
            const int wdt = 10 - m, cnt = (9 - m) * wdt;
            for (int idx = lane; idx < cnt; idx += 64) {
                const int ii = idx / wdt, i = m + 1 + ii, j = m + (idx - ii * wdt);
                const double y = a[i][m - 1] / x;
                a[i][j] -= y * a[m][j];
                if (j == m) ys[i] = y;
            }
            __syncthreads();
            // column m += sum_i y_i column i (one row per lane); the eliminated entries of column m-1 become zero
            if (lane < 10) {
                double acc = a[lane][m];
                for (int i = m + 1; i < 10; ++i) acc += ys[i] * a[lane][i];
                a[lane][m] = acc;
                if (lane > m) a[lane][m - 1] = 0.0;
            }
            __syncthreads();
        }
    }
}

// ---- 6. one root per SIX lanes: (y, z) from the eigenvalue (every lane of the group, same numbers), then Gauss-Newton on the
// ten cubic constraints with the rows dealt over the group: part p takes rows 2p, 2p+1 (p < 5).  Partial J^T J / J^T r
// go through w.root[k].ps; every lane of the group adds them up in the same order, so the six stay in step.
__device__ __forceinline__ void root_candidates(FptWorkspace& w)
{
    const int lane = (int)threadIdx.x;
    const int k = lane / 6 < 10 ? lane / 6 : 9, part = lane - 6 * (lane / 6);
    const bool grouped = lane < 60;
    FptWorkspace::Root& rw = w.root[k];
    double x = 0.0, yv = 0.0, z = 0.0;
    bool live = grouped && fpt_root_start(w, k, &x, &yv, &z);
    if (grouped && part == 0) rw.valid = 0;
    for (int it = 0; it < 4; ++it) {
        if (__ballot(live) == 0ull) break;
        double part_sums[9] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
        if (live && part < 5) {
            double mono[20];
            fpt_monomials(x, yv, z, mono);
#pragma unroll 1
            for (int r = 2 * part; r < 2 * part + 2; ++r) fpt_polish_row(w.M0[r], mono, part_sums);
        }
        __syncthreads();
        if (live) {
            double* slot = rw.ps[part];
#pragma unroll
            for (int e = 0; e < 9; ++e) slot[e] = part_sums[e];
        }
        __syncthreads();
        if (live) {
            double s[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) s[e] = 0.0;
            const double* all = &rw.ps[0][0];
#pragma unroll
            for (int pp = 0; pp < 5; ++pp)
#pragma unroll
                for (int e = 0; e < 9; ++e) s[e] += all[9 * pp + e];
            const int st = fpt_polish_update(s, &x, &yv, &z);
            if (st != 0) live = false;                            // converged or singular: this root's lanes leave together
            if (st < 0) { /* keeps the last iterate, like the sequential form */ }
        }
    }
    __syncthreads();
    if (grouped && part == 0 && fpt_root_is_real(w, k)) fpt_root_finish(w, k, x, yv, z);
    __syncthreads();
}

// ---- 7. compaction with the sequential rule of fpt_compact (a root is dropped if it repeats an ACCEPTED earlier one), the 45
// pairwise comparisons done by 45 lanes at once and the accept/reject chain on the resulting bit masks
__device__ __forceinline__ int compact(FptWorkspace& w, double* E_out)
{
    const int lane = (int)threadIdx.x;
    // pair index -> (k, s2), s2 < k: lane = k (k - 1) / 2 + s2
    int k = 1, s2 = 0;
    {
        int l = lane < 45 ? lane : 44;
        k = 1;
        while (l >= k) { l -= k; ++k; }
        s2 = l;
    }
    bool close = false;
    if (lane < 45 && w.root[k].valid && w.root[s2].valid) {
        const double* a = w.root[k].cand;
        const double* b = w.root[s2].cand;
        double scale = 0.0, d = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) { scale = fmax(scale, fabs(a[c])); d = fmax(d, fabs(a[c] - b[c])); }
        close = d < 1e-9 * (1.0 + scale);
    }
    const unsigned long long close_mask = __ballot(close);
    unsigned valid_mask = 0;
    for (int r = 0; r < 10; ++r) valid_mask |= w.root[r].valid ? (1u << r) : 0u;
    unsigned accepted = 0;
    for (int r = 0; r < 10; ++r) {
        if (!((valid_mask >> r) & 1u)) continue;
        const unsigned row_bits = (unsigned)((close_mask >> (r * (r - 1) / 2)) & ((1ull << r) - 1ull));     // bit s2: close to root s2
        if (row_bits & accepted) continue;
        accepted |= 1u << r;
    }
    __syncthreads();
    if (lane < 10 && ((accepted >> lane) & 1u)) {
        const int slot = __popc(accepted & ((1u << lane) - 1u));
#pragma unroll
        for (int c = 0; c < 9; ++c) E_out[9 * slot + c] = w.root[lane].cand[c];
    }
    __syncthreads();
    return __popc(accepted);
}

// q1, q2: 5 x 2 normalised coordinates (every lane holds the same values).  E_out (LDS): up to 10 x 9.
__device__ __forceinline__ int solve(const double q1[5][2], const double q2[5][2], double* E_out, FptWorkspace& w)
{
    const int lane = (int)threadIdx.x;
    // ---- 1
    double ee[4];
    if (!null_space(q1, q2, ee)) return 0;
    __syncthreads();
    if (lane < 9) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) w.EE[nb][lane] = ee[nb];
    }
    __syncthreads();
    FPT_STAMP(1);
    // ---- 2: lanes 0..8 build G = E E^T, lane 9 det(E); then lanes 0..8 the rows of 2 G E - tr(G) E
    const double (&EE)[4][9] = w.EE;
    double row[20];
#pragma unroll
    for (int mth = 0; mth < 20; ++mth) row[mth] = 0.0;
    auto load_e = [&](const int kk, double (&e)[20]) {
#pragma unroll
        for (int mth = 0; mth < 16; ++mth) e[mth] = 0.0;
        e[16] = EE[0][kk]; e[17] = EE[1][kk]; e[18] = EE[2][kk]; e[19] = EE[3][kk];
    };
    if (lane < 9) {
        const int a = lane / 3, b = lane - 3 * a;
        double g[20];
#pragma unroll
        for (int mth = 0; mth < 20; ++mth) g[mth] = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { double ea[20], eb[20]; load_e(3 * a + c, ea); load_e(3 * b + c, eb); fpt_mul_lin_lin(ea, eb, 1.0, g); }
#pragma unroll
        for (int mth = 10; mth < 20; ++mth) w.G[lane][mth] = g[mth];
    } else if (lane == 9) {
        const int tri[3][5] = { { 0, 4, 8, 5, 7 }, { 1, 3, 8, 5, 6 }, { 2, 3, 7, 4, 6 } };
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            double t[20], e0[20], e1[20], e2[20], e3[20], e4[20];
#pragma unroll
            for (int mth = 0; mth < 20; ++mth) t[mth] = 0.0;
            load_e(tri[s][0], e0); load_e(tri[s][1], e1); load_e(tri[s][2], e2); load_e(tri[s][3], e3); load_e(tri[s][4], e4);
            fpt_mul_lin_lin(e1, e2, 1.0, t);
            fpt_mul_lin_lin(e3, e4, -1.0, t);
            fpt_mul_quad_lin(t, e0, s == 1 ? -1.0 : 1.0, row);
        }
    }
    __syncthreads();
    if (lane < 9) {
        const int a = lane / 3, b = lane - 3 * a;
        double tr[20], eab[20];
#pragma unroll
        for (int mth = 0; mth < 10; ++mth) tr[mth] = 0.0;
#pragma unroll
        for (int mth = 10; mth < 20; ++mth) tr[mth] = w.G[0][mth] + w.G[4][mth] + w.G[8][mth];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double g[20], ecb[20];
#pragma unroll
            for (int mth = 0; mth < 10; ++mth) g[mth] = 0.0;
#pragma unroll
            for (int mth = 10; mth < 20; ++mth) g[mth] = w.G[3 * a + c][mth];
            load_e(3 * c + b, ecb);
            fpt_mul_quad_lin(g, ecb, 2.0, row);
        }
        load_e(3 * a + b, eab);
        fpt_mul_quad_lin(tr, eab, -1.0, row);
    }
    if (lane < 10) {
        const int ridx = lane == 9 ? 0 : 1 + lane;              // det(E) is row 0, the trace constraints rows 1..9
        double nr = 0.0;
#pragma unroll
        for (int mth = 0; mth < 20; ++mth) nr = fabs(row[mth]) > nr ? fabs(row[mth]) : nr;
        nr = nr > 0.0 ? 1.0 / nr : 1.0;
#pragma unroll
        for (int mth = 0; mth < 20; ++mth) { w.M[ridx][mth] = row[mth]; w.M0[ridx][mth] = row[mth] * nr; }
    }
    __syncthreads();
    FPT_STAMP(2);
    // ---- 3
    double mcol[10];
    {
        const int kk = lane < 20 ? lane : 19;
#pragma unroll
        for (int r = 0; r < 10; ++r) mcol[r] = w.M[r][kk];
    }
    if (!gauss_jordan(mcol)) return 0;
    // action matrix of multiplication by x on b = [x2 xy xz y2 yz z2 x y z 1] straight into the Hessenberg workspace too
    if (lane >= 10 && lane < 20) {
#pragma unroll
        for (int r = 0; r < 6; ++r) { w.Ax[10 * r + lane - 10] = -mcol[r]; w.hr[r][lane - 10] = -mcol[r]; }
    }
    if (lane < 40) {
        const double v = (lane == 0 || lane == 11 || lane == 22 || lane == 36) ? 1.0 : 0.0;
        w.Ax[60 + lane] = v;
        w.hr[6 + lane / 10][lane - 10 * (lane / 10)] = v;
    }
    __syncthreads();
    FPT_STAMP(3);
    // ---- 4, 5
    hessenberg(w);
    FPT_STAMP(4);
    fpt_balance10(w);
    double anorm = 0.0;
    for (int r = 0; r < 10; ++r) for (int c = (r > 0 ? r - 1 : 0); c < 10; ++c) anorm += fabs(w.hr[r][c]);
    if (!(anorm > 0.0)) return 0;
    fpt_aberth10(w, w.zr, w.zi, anorm);
    FPT_STAMP(5);
    // ---- 6
    root_candidates(w);
    FPT_STAMP(6);
    return compact(w, E_out);
}

// pixel -> normalised camera plane for an upper-triangular K (row-major)
__device__ __forceinline__ void normalise_px(const double* __restrict__ K, double u, double v, double* out)
{
    const double yn = (v - K[5]) / K[4];
    out[0] = (u - K[2] - K[1] * yn) / K[0];
    out[1] = yn;
}

// The ten {F, E} model slots (18 doubles each, NaN where the problem has fewer solutions) of ONE five-point problem on correspondences
// ids[0..4] of (x1, x2: N x 2 pixels; K1, K2: 3 x 3 row-major), solved by the calling WAVE (a 64-thread workgroup: solve() uses the
// workgroup barrier).  NOT inlined on purpose (round 5): fivept_kernel (pnp.hip: the hypotheses the tests hand to the a-contrario oracle)
// and acr_solve5_kernel (acransac.hip: the a-contrario round's own solve) call this one body, so that the two produce the same bits by
// construction -- this file is compiled with `fp contract(fast)`, and code inlined into two kernels may be contracted differently in
// each (the p3p.h story).  A call costs nothing against the ~40 us of a solve.  The workspace lives in LDS of its own.
__device__ __noinline__ void models_of_sample(const double* __restrict__ x1, const double* __restrict__ x2, const double* __restrict__ K1,
                                              const double* __restrict__ K2, const int i0, const int i1, const int i2, const int i3, const int i4,
                                              const int N, double* __restrict__ out /* 180 doubles, global */)
{
    __shared__ FptWorkspace ws;      // one problem per wave: every lane reads and writes the same values
    __shared__ double E[90];
    const int ids[5] = { i0, i1, i2, i3, i4 };
    double q1[5][2], q2[5][2];
    bool ok = true;
    for (int p = 0; p < 5; ++p) {
        int i = ids[p];
        if (i < 0 || i >= N) { ok = false; i = 0; }
        normalise_px(K1, x1[2 * i], x1[2 * i + 1], q1[p]);
        normalise_px(K2, x2[2 * i], x2[2 * i + 1], q2[p]);
    }
    const int n = ok ? solve(q1, q2, E, ws) : 0;
    // F = K2^-T E K1^-1 for upper-triangular K = [fx s cx; 0 fy cy; 0 0 1]: K^-1 = [1/fx, -s/(fx fy), (s cy - cx fy)/(fx fy); 0, 1/fy, -cy/fy; 0 0 1].
    // Plain IEEE operations in source order, NOT contracted (the block below switches the file's `fp contract(fast)` off): the tests restate
    // this product on the host (tests/test_gpu_acransac.py _f_from_e) to hand the oracle the F the device scores with, bit for bit.
    // One model slot per lane (round 5: lane 0 used to write all ten, ~1 000 dependent instructions on one lane at the end of every solve).
    {
#pragma clang fp contract(off)
        const int k = (int)threadIdx.x;
        if (k >= 10) return;
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        if (k >= n) {
            for (int e = 0; e < 18; ++e) out[18 * k + e] = qnan;
            return;
        }
        double A1[9], A2[9];
        {
            const double* Ks[2] = { K1, K2 };
            double* As[2] = { A1, A2 };
            for (int w = 0; w < 2; ++w) {
                const double fx = Ks[w][0], sk = Ks[w][1], cx = Ks[w][2], fy = Ks[w][4], cy = Ks[w][5];
                double* A = As[w];
                A[0] = 1.0 / fx; A[1] = -sk / (fx * fy); A[2] = (sk * cy - cx * fy) / (fx * fy);
                A[3] = 0.0; A[4] = 1.0 / fy; A[5] = -cy / fy;
                A[6] = 0.0; A[7] = 0.0; A[8] = 1.0;
            }
        }
        const double* Ek = E + 9 * k;
        double T[9];                                   // T = E K1^-1
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) T[3 * r + c] = Ek[3 * r] * A1[c] + Ek[3 * r + 1] * A1[3 + c] + Ek[3 * r + 2] * A1[6 + c];
        for (int r = 0; r < 3; ++r)                    // F = K2^-T T : F[r][c] = sum_m A2[m][r] T[m][c]
            for (int c = 0; c < 3; ++c) out[18 * k + 3 * r + c] = A2[r] * T[c] + A2[3 + r] * T[3 + c] + A2[6 + r] * T[6 + c];
        for (int e = 0; e < 9; ++e) out[18 * k + 9 + e] = Ek[e];
    }
}

} // namespace fpw

#endif // device
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#endif
