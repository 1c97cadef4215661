// Host build of the P3P statement (coloc_amd/csrc/p3p.h) as a tiny shared library for the tests: minimal solutions for the a-contrario
// oracle that do NOT come from the GPU (tests/test_gpu_acransac.py::test_pose_against_host_solved_oracle).  Same formulas as the
// device's p3p_sample_root -- bearings from K, p3p_prepare, p3p_pose_from_root per root -- compiled by g++ with IEEE division in
// place of v_rcp_f64 + Newton and the host compiler's own contraction choices, so results agree with the device to rounding, not
// bit for bit.  Test infrastructure only.
#include <cmath>
#include "../../coloc_amd/csrc/p3p.h"

// X: N x 3, x: N x 2, K: 9 row-major; sample: 3 indices; out: 4 pose slots of 12 doubles (slot = root, NaN when the root has no pose).
// Returns the number of valid slots.
extern "C" int p3p_host_sample(const double* X, const double* x, const double* K, const int* sample, double* out)
{
    double Xs[3][3], f[3][3];
    const double fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    for (int p = 0; p < 3; ++p) {
        const int i = sample[p];
        Xs[p][0] = X[3 * i]; Xs[p][1] = X[3 * i + 1]; Xs[p][2] = X[3 * i + 2];
        const double yn = (x[2 * i + 1] - cy) / fy;
        const double xn = (x[2 * i] - cx - sk * yn) / fx;
        const double inrm = 1.0 / std::sqrt(xn * xn + yn * yn + 1.0);
        f[p][0] = xn * inrm; f[p][1] = yn * inrm; f[p][2] = inrm;
    }
    P3PProblem prob;
    const bool ok = p3p_prepare(Xs, f, prob);
    int n = 0;
    for (int k = 0; k < 4; ++k) {
        double P[12];
        const bool have = ok && p3p_pose_from_root(prob, Xs, f, k, P);
        for (int e = 0; e < 12; ++e) out[12 * k + e] = have ? P[e] : NAN;
        n += have ? 1 : 0;
    }
    return n;
}
