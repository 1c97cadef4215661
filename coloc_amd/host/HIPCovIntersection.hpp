// HIPCovIntersection.hpp -- dependency-free stand-in for coloc's CovIntersection (reference
// include/coloc/CovIntersection.hpp:11-70), the 3x3 fusion step that closes the inter-camera loop
// (include/coloc/coloc.hpp:362-389).  Host arithmetic only (3x3 algebra) -- kept
// beside the GPU policy classes so that the streaming loop needs neither dlib nor OpenMVG types.
//
// Same members and meaning:
//   loadData(CA, CB, ca, cb)       covariances / positions of the two estimates
//   optimize()                      omega in [0, 1] minimising trace(C(omega)),
//                                   C(omega) = inv(inv(CA) + inv(CB) - inv(omega CA + (1 - omega) CB))      (:24-29)
//                                   dlib::find_min_single_variable(start 0.0, [0,1], eps 1e-3, 100 it.)   (:34-38, :58-63)
//   computeFusedValues()            covFused = C(minX); poseFused = K ca + L cb with
//                                   K = C (inv(CA) - minX M), L = C (inv(CB) - (1 - minX) M), M = inv(minX CA + (1-minX) CB)  (:40-49)
// dlib is an empty, unpinned submodule in the reference snapshot, so the exact iterates of its 1-D
// search cannot be reproduced; this uses a bounded golden-section search to the same tolerance
// (|omega - omega*| <= 1e-3), which bounds the difference of the fused outputs accordingly.
#pragma once

#include <array>
#include <cmath>

namespace coloc {

using Mat3d = std::array<double, 9>;    // row-major
using Vec3d = std::array<double, 3>;

namespace ci_detail {
inline Mat3d inv3(const Mat3d& m)
{
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double r = 1.0 / det;
    return { A * r, -(b * i - c * h) * r, (b * f - c * e) * r,
             B * r, (a * i - c * g) * r, -(a * f - c * d) * r,
             C * r, -(a * h - b * g) * r, (a * e - b * d) * r };
}
inline Mat3d add(const Mat3d& x, const Mat3d& y, double sy = 1.0)
{
    Mat3d r;
    for (int k = 0; k < 9; ++k) r[k] = x[k] + sy * y[k];
    return r;
}
inline Mat3d scale(const Mat3d& x, double s)
{
    Mat3d r;
    for (int k = 0; k < 9; ++k) r[k] = s * x[k];
    return r;
}
inline Mat3d mul(const Mat3d& x, const Mat3d& y)
{
    Mat3d r{};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) r[3 * i + j] += x[3 * i + k] * y[3 * k + j];
    return r;
}
inline Vec3d mul(const Mat3d& x, const Vec3d& v)
{
    return { x[0] * v[0] + x[1] * v[1] + x[2] * v[2], x[3] * v[0] + x[4] * v[1] + x[5] * v[2], x[6] * v[0] + x[7] * v[1] + x[8] * v[2] };
}
} // namespace ci_detail

class HIPCovIntersection {
public:
    void loadData(const Mat3d& covA, const Mat3d& covB, const Vec3d& posA, const Vec3d& posB)
    {
        CA = covA; CB = covB; ca = posA; cb = posB;
    }

    // CovIntersection::function (:24-29)
    double function(double x) const
    {
        const Mat3d c = fused(x);
        return c[0] + c[4] + c[8];
    }

    void optimize()
    {
        // golden-section search on [begin, end] down to an interval of eps; end points are candidates too
        const double gr = 0.6180339887498949;
        double a = begin, b = end;
        double x1 = b - gr * (b - a), x2 = a + gr * (b - a);
        double f1 = function(x1), f2 = function(x2);
        for (long it = 0; it < max_iter && (b - a) > eps; ++it) {
            if (f1 < f2) { b = x2; x2 = x1; f2 = f1; x1 = b - gr * (b - a); f1 = function(x1); }
            else { a = x1; x1 = x2; f1 = f2; x2 = a + gr * (b - a); f2 = function(x2); }
        }
        double bx = 0.5 * (a + b), bf = function(bx);
        const double fb = function(begin), fe = function(end);
        if (fb < bf) { bf = fb; bx = begin; }
        if (fe < bf) { bf = fe; bx = end; }
        minX = bx;
        minValue = bf;
    }

    void computeFusedValues()
    {
        using namespace ci_detail;
        covFused = fused(minX);
        const Mat3d M = inv3(add(scale(CA, minX), scale(CB, 1.0 - minX)));
        const Mat3d KICI = mul(covFused, add(inv3(CA), M, -minX));
        const Mat3d LICI = mul(covFused, add(inv3(CB), M, -(1.0 - minX)));
        const Vec3d pa = mul(KICI, ca), pb = mul(LICI, cb);
        poseFused = { pa[0] + pb[0], pa[1] + pb[1], pa[2] + pb[2] };
    }

    // translation block of a row-major 6x6 [angle-axis | t] covariance: indices 21..23, 27..29, 33..35
    // (coloc::Utils::loadPoseCovariance, include/coloc/colocUtils.hpp:38-43)
    static Mat3d translationBlock(const std::array<double, 36>& cov)
    {
        return { cov[21], cov[22], cov[23], cov[27], cov[28], cov[29], cov[33], cov[34], cov[35] };
    }

    double minValue = 0.0;
    double minX = 0.0;
    Mat3d covFused{};
    Vec3d poseFused{};

private:
    Mat3d fused(double x) const
    {
        using namespace ci_detail;
        return inv3(add(add(inv3(CA), inv3(CB)), inv3(add(scale(CA, x), scale(CB, 1.0 - x))), -1.0));
    }
    Mat3d CA{}, CB{};
    Vec3d ca{}, cb{};
    const double begin = 0.0, end = 1.0, eps = 1e-3;
    const long max_iter = 100;
};

} // namespace coloc
