// coloc_hip_geometry.hpp -- the geometric types HIPLocalizer / HIPRobustMatcher exchange with CoLoC.
//
// Inside the reference tree (COLOC_HIP_WITH_OPENMVG defined) the real openMVG / Eigen types are used and this file
// adds nothing.  Stand-alone (this repository: OpenMVG and Eigen are not installed) the minimal stand-ins below carry
// exactly the members the two policy classes touch, with the same names and meaning:
//   openMVG::Vec2 / Vec3 / Mat3 (fixed size, operator()(r, c) / operator()(i) / operator[]), openMVG::Mat (dynamic,
//   column-major like Eigen: resize(rows, cols), operator()(r, c), rows(), cols()),
//   openMVG::geometry::Pose3 (rotation(), center(), translation(); Pose3(R, C)),
//   openMVG::cameras::Pinhole_Intrinsic_Radial_K3 (ctor (w, h, focal, ppx, ppy, k1, k2, k3), have_disto(),
//   get_ud_pixel() with OpenMVG's bisection on r^2 (1 + k1 r^2 + k2 r^4 + k3 r^6)^2, operator()(2 x N pixels) -> 3 x N bearings, K(), Kinv()),
//   openMVG::sfm::Image_Localizer_Match_Data (pt3D, pt2D, vec_inliers, error_max, max_iteration),
//   openMVG::sfm::Landmark{X} / Landmarks and the slice of SfM_Data ("Scene") Localizer::setupTracks reads,
//   coloc::Cov6 (6 x 6 pose covariance, [angle-axis | translation] order as PoseRefiner::refinePose fills it,
//   reference include/coloc/Refiner.hpp:177-197), coloc::colocParams (include/coloc/colocParams.hpp:19-37),
//   openMVG::sfm::RelativePose_Info as RobustMatcher uses it (essential_matrix, vec_inliers, relativePose,
//   initial_residual_tolerance, found_residual_precision).
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "coloc_hip_types.hpp"

#ifndef COLOC_HIP_WITH_OPENMVG

namespace openMVG {

struct Mat3 {
    std::array<double, 9> m{};                         // row-major storage; access only through (r, c)
    double& operator()(size_t r, size_t c) { return m[3 * r + c]; }
    double operator()(size_t r, size_t c) const { return m[3 * r + c]; }
    static Mat3 Identity() { Mat3 I; I(0, 0) = I(1, 1) = I(2, 2) = 1.0; return I; }
};

// dynamic matrix, column-major (Eigen's default): pt3D is 3 x N, pt2D is 2 x N (Localizer.hpp:61-62)
class Mat {
public:
    Mat() = default;
    Mat(size_t r, size_t c) { resize(r, c); }
    void resize(size_t r, size_t c) { rows_ = r; cols_ = c; d_.assign(r * c, 0.0); }
    double& operator()(size_t r, size_t c) { return d_[c * rows_ + r]; }
    double operator()(size_t r, size_t c) const { return d_[c * rows_ + r]; }
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    const double* data() const { return d_.data(); }
private:
    size_t rows_ = 0, cols_ = 0;
    std::vector<double> d_;
};
using Mat2X = Mat;
using Mat3X = Mat;

namespace geometry {
// x_cam = R (X - C)
class Pose3 {
public:
    Pose3() : R_(Mat3::Identity()) {}
    Pose3(const Mat3& R, const Vec3& C) : R_(R), C_(C) {}
    const Mat3& rotation() const { return R_; }
    Mat3& rotation() { return R_; }
    const Vec3& center() const { return C_; }
    Vec3& center() { return C_; }
    Vec3 translation() const
    {
        Vec3 t;
        for (int i = 0; i < 3; ++i) t[i] = -(R_(i, 0) * C_[0] + R_(i, 1) * C_[1] + R_(i, 2) * C_[2]);
        return t;
    }
private:
    Mat3 R_;
    Vec3 C_;
};
} // namespace geometry

namespace cameras {
// pinhole camera with 3 radial distortion coefficients; (focal, ppx, ppy) in pixels
class Pinhole_Intrinsic_Radial_K3 {
public:
    Pinhole_Intrinsic_Radial_K3(int w = 0, int h = 0, double focal = 0.0, double ppx = 0.0, double ppy = 0.0, double k1 = 0.0,
                                double k2 = 0.0, double k3 = 0.0)
        : w_(w), h_(h), f_(focal), ppx_(ppx), ppy_(ppy), k_{ { k1, k2, k3 } } {}
    bool have_disto() const { return true; }
    double focal() const { return f_; }
    Vec2 principal_point() const { return Vec2(ppx_, ppy_); }
    Mat3 K() const { Mat3 K; K(0, 0) = f_; K(1, 1) = f_; K(0, 2) = ppx_; K(1, 2) = ppy_; K(2, 2) = 1.0; return K; }
    Mat3 Kinv() const { Mat3 Ki; Ki(0, 0) = 1.0 / f_; Ki(1, 1) = 1.0 / f_; Ki(0, 2) = -ppx_ / f_; Ki(1, 2) = -ppy_ / f_; Ki(2, 2) = 1.0; return Ki; }
    Vec2 ima2cam(const Vec2& p) const { return Vec2((p[0] - ppx_) / f_, (p[1] - ppy_) / f_); }
    Vec2 cam2ima(const Vec2& p) const { return Vec2(f_ * p[0] + ppx_, f_ * p[1] + ppy_); }
    double imagePlane_toCameraPlaneError(double v) const { return v / f_; }
    // undistorted pixel of a distorted one: radius found by bisection on the (monotone) distortion function
    Vec2 get_ud_pixel(const Vec2& p) const
    {
        const Vec2 c = ima2cam(p);
        const double r2 = c[0] * c[0] + c[1] * c[1];
        const double radius = (r2 == 0.0) ? 1.0 : std::sqrt(bisection_radius_solve(r2) / r2);
        return cam2ima(Vec2(radius * c[0], radius * c[1]));
    }
    // bearing vectors of (undistorted) pixels, 2 x N -> 3 x N: (x, y, 1) on the camera plane, normalised -- the member the
    // reference calls, IntrinsicBase::operator()(const Mat2X&) (RobustMatcher.hpp:159)
    Mat operator()(const Mat& p) const
    {
        Mat b(3, p.cols());
        for (size_t i = 0; i < p.cols(); ++i) {
            const Vec2 c = ima2cam(Vec2(p(0, i), p(1, i)));
            const double n = std::sqrt(c[0] * c[0] + c[1] * c[1] + 1.0);
            b(0, i) = c[0] / n; b(1, i) = c[1] / n; b(2, i) = 1.0 / n;
        }
        return b;
    }
    int w() const { return w_; }
    int h() const { return h_; }
private:
    double disto_functor(double r2) const
    {
        const double t = 1.0 + r2 * (k_[0] + r2 * (k_[1] + r2 * k_[2]));
        return r2 * t * t;
    }
    double bisection_radius_solve(double r2, double epsilon = 1e-10) const
    {
        double lowerbound = r2, upbound = r2;
        while (disto_functor(lowerbound) > r2) lowerbound /= 1.05;
        while (disto_functor(upbound) < r2) upbound *= 1.05;
        while (epsilon < upbound - lowerbound) {
            const double mid = .5 * (lowerbound + upbound);
            if (disto_functor(mid) > r2) upbound = mid;
            else lowerbound = mid;
        }
        return .5 * (lowerbound + upbound);
    }
    int w_, h_;
    double f_, ppx_, ppy_;
    std::array<double, 3> k_;
};
using IntrinsicBase = Pinhole_Intrinsic_Radial_K3;
using Pinhole_Intrinsic = Pinhole_Intrinsic_Radial_K3;
} // namespace cameras

namespace sfm {
struct Image_Localizer_Match_Data {
    Mat pt3D, pt2D;                                   // 3 x N, 2 x N
    std::vector<uint32_t> vec_inliers;
    double error_max = std::numeric_limits<double>::infinity();
    size_t max_iteration = 4096;
};
struct RelativePose_Info {
    Mat3 essential_matrix;
    std::vector<uint32_t> vec_inliers;
    geometry::Pose3 relativePose;
    double initial_residual_tolerance = std::numeric_limits<double>::infinity();
    double found_residual_precision = std::numeric_limits<double>::infinity();
};
} // namespace sfm
} // namespace openMVG

namespace coloc {
struct Cov6 {                                          // Eigen::Matrix<double, 6, 6> in the reference
    std::array<double, 36> m{};
    double& operator()(size_t r, size_t c) { return m[6 * r + c]; }
    double operator()(size_t r, size_t c) const { return m[6 * r + c]; }
    static Cov6 Identity() { Cov6 I; for (int i = 0; i < 6; ++i) I(i, i) = 1.0; return I; }
};

// include/coloc/colocParams.hpp:19-37
class colocParams {
public:
    std::string imageFolder;
    std::pair<int, int> imageSize;
    std::vector<openMVG::Mat3> K;
    std::vector<openMVG::Vec3> dist;
    char model = 'E';
    DetectorOptions detectorOptions{};
    MatcherOptions matcherOptions{};
    colocParams() = default;
    colocParams(const std::vector<openMVG::Mat3>& _K, const std::vector<openMVG::Vec3>& _dist, const char& _model,
                const std::pair<size_t, size_t>& _imageSize, const std::string& _imageFolder, DetectorOptions _detectorOptions,
                MatcherOptions _matcherOptions)
        : imageFolder(_imageFolder), imageSize(static_cast<int>(_imageSize.first), static_cast<int>(_imageSize.second)), K(_K),
          dist(_dist), model(_model), detectorOptions(_detectorOptions), matcherOptions(_matcherOptions) {}
};

} // namespace coloc

#endif // COLOC_HIP_WITH_OPENMVG
