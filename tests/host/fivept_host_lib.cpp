// Host build of the sequential five-point statement (coloc_amd/csrc/fivept.h) as a tiny shared library for the tests: the
// checker for the wave-cooperative GPU form (coloc_amd/csrc/fivept_wave.h).  Test infrastructure only.
#include "../../coloc_amd/csrc/fivept.h"

extern "C" int fpt_host_solve(const double* q1, const double* q2, double* E_out /* 90 */)
{
    static thread_local FptWorkspace ws;
    double a[5][2], b[5][2];
    for (int p = 0; p < 5; ++p) for (int c = 0; c < 2; ++c) { a[p][c] = q1[2 * p + c]; b[p][c] = q2[2 * p + c]; }
    const int n = fivept_solve(a, b, E_out, ws);
    for (int i = 9 * n; i < 90; ++i) E_out[i] = 0.0;
    return n;
}
