// Host build of the PRODUCT's seven-point / four-point statements (coloc_amd/csrc/twoview_min.h) as a tiny shared library for the
// tests: they are held against numpy's SVD and against the oracle's own solvers (oracle/clc_oracle_twoview.c).  Test infrastructure only.
#include "../../coloc_amd/csrc/twoview_min.h"

extern "C" int tv_host_seven_point(const double* q1, const double* q2, double* F_out /* 27 */)
{
    double a[7][2], b[7][2], F[3][9];
    for (int p = 0; p < 7; ++p) for (int c = 0; c < 2; ++c) { a[p][c] = q1[2 * p + c]; b[p][c] = q2[2 * p + c]; }
    const int n = clc::tv::seven_point(a, b, F);
    for (int k = 0; k < 3; ++k) for (int e = 0; e < 9; ++e) F_out[9 * k + e] = k < n ? F[k][e] : 0.0;
    return n;
}
extern "C" int tv_host_four_point(const double* q1, const double* q2, double* H_out /* 9 */)
{
    double a[4][2], b[4][2], H[9];
    for (int p = 0; p < 4; ++p) for (int c = 0; c < 2; ++c) { a[p][c] = q1[2 * p + c]; b[p][c] = q2[2 * p + c]; }
    const int n = clc::tv::four_point(a, b, H);
    for (int e = 0; e < 9; ++e) H_out[e] = H[e];
    return n;
}
extern "C" void tv_host_normalizer(int w, int h, double* t3 /* d, tx, ty */)
{
    const clc::tv::Normalizer t = clc::tv::normalizer(w, h);
    t3[0] = t.d; t3[1] = t.tx; t3[2] = t.ty;
}
extern "C" void tv_host_unnormalize(int homography, int w, int h, const double* Mn, double* M)
{
    clc::tv::unnormalize(homography != 0, clc::tv::normalizer(w, h), Mn, M);
}
extern "C" int tv_host_cubic(double a, double b, double c, double* x)
{
    double r[3];
    const int n = clc::tv::cubic_roots(a, b, c, r);
    for (int k = 0; k < 3; ++k) x[k] = r[k];
    return n;
}
