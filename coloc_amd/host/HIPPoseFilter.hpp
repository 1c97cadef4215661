// HIPPoseFilter.hpp -- dependency-free stand-in for coloc::colocFilter (reference
// include/coloc/KalmanFilter.hpp:8-165), the per-drone 6-state Kalman filter with the chi-square gate
// that smooths the poses coming out of the PnP stage (include/coloc/coloc.hpp:230-262).  Host
// arithmetic only (6x6 algebra): it sits beside the GPU policy classes so that the
// streaming loop needs neither OpenCV nor OpenMVG types.
//
// The reference builds on cv::KalmanFilter (OpenCV is absent from the snapshot; the filter equations
// are the textbook ones cv::KalmanFilter documents).  What the reference configures, and this keeps:
//   state = measurement = [x y z roll pitch yaw], no control input                         (:109-113)
//   transition = I (cv::KalmanFilter::init default), measurement matrix = I               (:123-128)
//   process noise 1e-2 I, measurement noise 1e-1 I, posterior error covariance I, state 0 (:118-120)
//   update(): predict; measurement noise rows/cols 3..5 <- cov[21..23,27..29,33..35] * rmse (:46-58);
//             if a measurement was filled: gate, then correct unless the gate rejected and the filter
//             has left its initial phase; pose <- (euler2rot(state[3..5]), state[0..2])      (:60-93)
//   gate: d = innovation^T * S * innovation with S = H P' H^T + R -- the reference multiplies by S,
//         NOT by its inverse (:148) -- reject when d > 10 (:161)
//   the initial phase ends after the first update of drone 2 (:95-96, a hard-coded id; kept as the
//         default of `initEndsAtDrone`)
//   one `measurementsAvailable` flag shared by all drones (:111), cleared by every update (:94)
// Differences: the flag starts false (uninitialised in the reference); the gate value is returned in
// `lastGateDistance` instead of being printed and appended to ./mahalanobis.txt (:153-159).
//
// rot2euler / euler2rot follow include/coloc/colocUtils.hpp:63-140 (bank / attitude / heading with the
// +-0.998 pole cut).
#pragma once

#include <array>
#include <cmath>
#include <vector>

namespace coloc {

using Mat6d = std::array<double, 36>;   // row-major
using Vec6d = std::array<double, 6>;
using Cov6d = std::array<double, 36>;   // same layout as coloc::Cov6 (colocData.hpp): row-major 6x6, [rotation | translation]... rows 3..5 = indices 21..35

namespace kf_detail {
inline Mat6d identity(double s)
{
    Mat6d m{};
    for (int i = 0; i < 6; ++i) m[7 * i] = s;
    return m;
}
// solve S X = B for a symmetric positive definite 6x6 S by Gauss-Jordan with partial pivoting (B has 6 columns)
inline bool solve6(Mat6d S, Mat6d B, Mat6d& X)
{
    for (int c = 0; c < 6; ++c) {
        int p = c;
        for (int r = c + 1; r < 6; ++r)
            if (std::fabs(S[6 * r + c]) > std::fabs(S[6 * p + c])) p = r;
        if (!(std::fabs(S[6 * p + c]) > 0.0)) return false;
        if (p != c)
            for (int k = 0; k < 6; ++k) { std::swap(S[6 * p + k], S[6 * c + k]); std::swap(B[6 * p + k], B[6 * c + k]); }
        const double inv = 1.0 / S[6 * c + c];
        for (int k = 0; k < 6; ++k) { S[6 * c + k] *= inv; B[6 * c + k] *= inv; }
        for (int r = 0; r < 6; ++r) {
            if (r == c) continue;
            const double f = S[6 * r + c];
            if (f == 0.0) continue;
            for (int k = 0; k < 6; ++k) { S[6 * r + k] -= f * S[6 * c + k]; B[6 * r + k] -= f * B[6 * c + k]; }
        }
    }
    X = B;
    return true;
}
} // namespace kf_detail

// colocUtils.hpp:63-100.  R row-major 3x3 -> {bank, attitude, heading}
inline std::array<double, 3> rot2euler(const std::array<double, 9>& R)
{
    const double m00 = R[0], m02 = R[2], m10 = R[3], m11 = R[4], m12 = R[5], m20 = R[6], m22 = R[8];
    const double half_pi = 1.5707963267948966;
    if (m10 > 0.998) return { 0.0, half_pi, std::atan2(m02, m22) };
    if (m10 < -0.998) return { 0.0, -half_pi, std::atan2(m02, m22) };
    return { std::atan2(-m12, m11), std::asin(m10), std::atan2(-m20, m00) };
}

// colocUtils.hpp:102-141
inline std::array<double, 9> euler2rot(const std::array<double, 3>& e)
{
    const double ch = std::cos(e[2]), sh = std::sin(e[2]);
    const double ca = std::cos(e[1]), sa = std::sin(e[1]);
    const double cb = std::cos(e[0]), sb = std::sin(e[0]);
    return { ch * ca, sh * sb - ch * sa * cb, ch * sa * sb + sh * cb,
             sa, ca * cb, -ca * sb,
             -sh * ca, sh * sa * cb + ch * sb, -sh * sa * sb + ch * cb };
}

class HIPPoseFilter {
public:
    struct Filter {
        Vec6d statePre{}, statePost{};
        Mat6d errorCovPre{}, errorCovPost = kf_detail::identity(1.0);
        Mat6d processNoiseCov = kf_detail::identity(1e-2);
        Mat6d measurementNoiseCov = kf_detail::identity(1e-1);
    };
    std::vector<Filter> droneFilters;
    std::vector<Vec6d> droneMeasurements;
    double lastGateDistance = 0.0;      // d of the most recent gate evaluation
    bool lastRejected = false;          // the most recent update kept the prediction because the gate rejected
    double gateThreshold = 10.0;        // KalmanFilter.hpp:161
    int initEndsAtDrone = 2;            // KalmanFilter.hpp:95

    explicit HIPPoseFilter(unsigned int nDrones) : droneFilters(nDrones), droneMeasurements(nDrones, Vec6d{}) {}

    // KalmanFilter.hpp:24-42.  rotation row-major.
    void fillMeasurements(Vec6d& measurements, const std::array<double, 3>& translation, const std::array<double, 9>& rotation)
    {
        const std::array<double, 3> e = rot2euler(rotation);
        measurements = { translation[0], translation[1], translation[2], e[0], e[1], e[2] };
        measurementsAvailable = true;
    }

    // KalmanFilter.hpp:44-97.  Outputs the filtered pose as (R row-major, t).
    void update(int droneId, const Cov6d& cov, float rmse, std::array<double, 9>& R_out, std::array<double, 3>& t_out)
    {
        Filter& f = droneFilters[(size_t)droneId];
        // predict (transition = I): x' = x, P' = P + Q; cv::KalmanFilter also copies them into the posterior
        f.statePre = f.statePost;
        for (int k = 0; k < 36; ++k) f.errorCovPre[k] = f.errorCovPost[k] + f.processNoiseCov[k];
        f.errorCovPost = f.errorCovPre;
        const Vec6d predicted = f.statePre;

        static const int src[9] = { 21, 22, 23, 27, 28, 29, 33, 34, 35 };
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) f.measurementNoiseCov[6 * (3 + r) + 3 + c] = cov[(size_t)src[3 * r + c]] * (double)rmse;

        Vec6d estimated = predicted;
        lastRejected = false;
        if (measurementsAvailable) {
            const bool reject = chiSquareGating(f, droneMeasurements[(size_t)droneId], predicted);
            if (reject && !init) {
                lastRejected = true;
            } else {
                estimated = correct(f, droneMeasurements[(size_t)droneId]);
            }
        }
        t_out = { estimated[0], estimated[1], estimated[2] };
        R_out = euler2rot({ estimated[3], estimated[4], estimated[5] });
        measurementsAvailable = false;
        if (droneId == initEndsAtDrone) init = false;
    }

    bool inInitialPhase() const { return init; }

private:
    bool init = true;
    bool measurementsAvailable = false;

    // S = H P' H^T + R with H = I
    static Mat6d innovationCov(const Filter& f)
    {
        Mat6d S;
        for (int k = 0; k < 36; ++k) S[k] = f.errorCovPre[k] + f.measurementNoiseCov[k];
        return S;
    }

    // KalmanFilter.hpp:130-164
    bool chiSquareGating(const Filter& f, const Vec6d& z, const Vec6d& predicted)
    {
        Vec6d innv;
        for (int i = 0; i < 6; ++i) innv[i] = z[i] - predicted[i];
        const Mat6d S = innovationCov(f);
        double d = 0.0;
        for (int i = 0; i < 6; ++i) {
            double row = 0.0;
            for (int j = 0; j < 6; ++j) row += S[6 * i + j] * innv[j];
            d += innv[i] * row;
        }
        lastGateDistance = d;
        return d > gateThreshold;
    }

    // cv::KalmanFilter::correct with H = I: K = P' S^-1, x = x' + K (z - x'), P = P' - K P'
    static Vec6d correct(Filter& f, const Vec6d& z)
    {
        const Mat6d S = innovationCov(f);
        // K^T = S^-1 P'^T  (S symmetric)  ->  solve S Kt = P'^T
        Mat6d PT, Kt;
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) PT[6 * i + j] = f.errorCovPre[6 * j + i];
        if (!kf_detail::solve6(S, PT, Kt)) { f.statePost = f.statePre; return f.statePost; }
        Vec6d y;
        for (int i = 0; i < 6; ++i) y[i] = z[i] - f.statePre[i];
        for (int i = 0; i < 6; ++i) {
            double acc = 0.0;
            for (int j = 0; j < 6; ++j) acc += Kt[6 * j + i] * y[j];
            f.statePost[i] = f.statePre[i] + acc;
        }
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 6; ++k) acc += Kt[6 * k + i] * f.errorCovPre[6 * k + j];
                f.errorCovPost[6 * i + j] = f.errorCovPre[6 * i + j] - acc;
            }
        return f.statePost;
    }
};

} // namespace coloc
