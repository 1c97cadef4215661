// HIPPoseLog.hpp -- the pose / covariance CSV wire format of coloc::Logger (reference
// include/coloc/logUtils.hpp:21-100), dependency-free.  Downstream plotting scripts
// parse these lines, so field order, separators and number formatting are kept:
//
//   idx,dest,source,cx,cy,cz,c21,c22,c23,c27,c28,c29,c33,c34,c35,roll,pitch,yaw,rmse,nTracks\n
//
//   * note the argument order of the reference call: logPoseCovtoFile(idx, source, dest, ...) writes
//     `dest` BEFORE `source` (:94)
//   * numbers go through a default-formatted std::ostream (6 significant digits, %g style)
//   * position = pose.center(); c.. = the translation block of the 6x6 covariance
//   * roll/pitch/yaw: Eigen eulerAngles(2, 1, 0) of the rotation, remapped by convertAnglesForLogging
//     (:36-67, evaluated in float as there) and printed in degrees as float
//
// logPosetoPLY / logMaptoPLY (:102-167): the ASCII PLY track and map dumps, same header and vertex lines.
//
// Eigen is absent from the snapshot (unpinned): eulerAngles(2,1,0) is restated from Eigen 3.3's published
// algorithm (first angle in [0, pi]); tests check R == Rz(a0) Ry(a1) Rx(a2) for the returned triple.
#pragma once

#include <array>
#include <cerrno>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <limits>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace coloc {

// Eigen 3.3 MatrixBase::eulerAngles(2, 1, 0) for a row-major rotation matrix
inline std::array<double, 3> eulerAnglesZYX(const std::array<double, 9>& R)
{
    auto m = [&](int r, int c) { return R[(size_t)(3 * r + c)]; };
    const double pi = 3.14159265358979323846;
    // a0 = 2, a1 = 1, a2 = 0  ->  odd = 1, i = 2, j = 1, k = 0
    const int i = 2, j = 1, k = 0;
    std::array<double, 3> res;
    res[0] = std::atan2(m(j, k), m(k, k));
    const double c2 = std::sqrt(m(i, i) * m(i, i) + m(i, j) * m(i, j));
    if (res[0] < 0.0) {                       // odd permutation: fold the first angle into [0, pi]
        res[0] += pi;
        res[1] = std::atan2(-m(i, k), -c2);
    } else {
        res[1] = std::atan2(-m(i, k), c2);
    }
    const double s1 = std::sin(res[0]), c1 = std::cos(res[0]);
    res[2] = std::atan2(s1 * m(k, i) - c1 * m(j, i), c1 * m(j, j) - s1 * m(k, j));
    return res;
}

// logUtils.hpp:36-67 (float arithmetic, as in the reference; `abs` there resolves to the float overload)
inline void convertAnglesForLogging(std::array<double, 3>& angles)
{
    const double pi = 3.14159265358979323846;
    float a1 = (float)(angles[0] * 180 / pi);
    float a2 = (float)(angles[2] * 180 / pi);
    float a3 = (float)(angles[1] * 180 / pi);
    if (std::fabs(a2) > 120) a2 = a2 < 0 ? (-1 * a2 - 180) : 180 - a2;
    if (std::fabs(a3) > 120) a3 = a3 < 0 ? 180 + a3 : a3 - 180;
    else a3 = -1 * a3;
    if (std::fabs(a1) > 120) a1 = a1 < 0 ? 180 + a1 : a1 - 180;
    angles[0] = a1 * pi / 180;
    angles[1] = a2 * pi / 180;
    angles[2] = a3 * pi / 180;
}

class HIPPoseLog {
public:
    // logUtils.hpp:24-34: truncate / create.  Returns EXIT_SUCCESS (0 == false) on success, as the reference does.
    bool createLogFile(const std::string& filename)
    {
        std::ofstream file;
        file.open(filename, std::ofstream::out | std::ofstream::trunc);
        file.close();
        return file.fail() ? true : false;
    }

    // one CSV record to any stream
    static void writePoseCov(std::ostream& os, int idx, int source, int dest, const std::array<double, 9>& rotation,
                             const std::array<double, 3>& center, const std::array<double, 36>& cov, float rmse, int nTracks)
    {
        const double pi = 3.14159265358979323846;
        std::array<double, 3> e = eulerAnglesZYX(rotation);
        convertAnglesForLogging(e);
        const float roll = (float)(e[0] * 180 / pi), pitch = (float)(e[1] * 180 / pi), yaw = (float)(e[2] * 180 / pi);
        os << idx << "," << dest << "," << source << ","
           << center[0] << "," << center[1] << "," << center[2] << ","
           << cov[21] << "," << cov[22] << "," << cov[23] << ","
           << cov[27] << "," << cov[28] << "," << cov[29] << ","
           << cov[33] << "," << cov[34] << "," << cov[35] << ","
           << roll << "," << pitch << "," << yaw << "," << rmse << "," << nTracks << std::endl;
    }

    // logUtils.hpp:69-100: append one record; throws std::ios_base::failure when the file cannot be opened or written
    bool logPoseCovtoFile(int idx, int source, int dest, const std::array<double, 9>& rotation, const std::array<double, 3>& center,
                          const std::array<double, 36>& cov, float rmse, int nTracks, const std::string& filename)
    {
        std::ofstream file;
        file.open(filename, std::ios::out | std::ios::app);
        if (file.fail()) throw std::ios_base::failure(std::strerror(errno));
        file.exceptions(file.exceptions() | std::ios::failbit | std::ifstream::badbit);
        writePoseCov(file, idx, source, dest, rotation, center, cov, rmse, nTracks);
        return file.good();
    }

    // logUtils.hpp:102-120: APPEND one vertex "cx cy cz 0 255 0" (fixed, digits10 + 1 decimals) -- the camera track ColoC collects
    // (coloc.hpp:185, 243).  Returns stream.good(): TRUE = written, the opposite of the EXIT_* convention above, as in the reference.
    bool logPosetoPLY(const std::array<double, 3>& center, const std::string& filename)
    {
        std::ofstream stream(filename.c_str(), std::ios::out | std::ios::app);
        if (!stream.is_open()) return false;
        stream << std::fixed << std::setprecision(std::numeric_limits<double>::digits10 + 1);
        stream << center[0] << ' ' << center[1] << ' ' << center[2] << ' ' << "0 255 0\n";
        return stream.good();
    }

    // logUtils.hpp:122-167: ASCII PLY of a map: header, the centre of every posed view in green, every landmark in white
    // (coloc.hpp:427).  The scene is handed over as its two point lists (Scene::GetPoses() centres, GetLandmarks() X).
    bool logMaptoPLY(const std::vector<std::array<double, 3>>& poseCenters, const std::vector<std::array<double, 3>>& landmarks,
                     const std::string& filename)
    {
        std::ofstream stream(filename.c_str(), std::ios::out | std::ios::binary);
        if (!stream.is_open()) return false;
        stream << std::fixed << std::setprecision(std::numeric_limits<double>::digits10 + 1);
        stream << "ply" << '\n' << "format " << "ascii 1.0"
               << '\n' << "comment generated by coloc"
               << '\n' << "element vertex " << landmarks.size() + poseCenters.size()
               << '\n' << "property double x" << '\n' << "property double y" << '\n' << "property double z"
               << '\n' << "property uchar red" << '\n' << "property uchar green" << '\n' << "property uchar blue"
               << '\n' << "end_header" << std::endl;
        for (const auto& c : poseCenters) stream << c[0] << ' ' << c[1] << ' ' << c[2] << ' ' << "0 255 0\n";
        for (const auto& X : landmarks) stream << X[0] << ' ' << X[1] << ' ' << X[2] << ' ' << "255 255 255\n";
        stream.flush();
        const bool logStatus = stream.good();
        stream.close();
        return logStatus;
    }
};

} // namespace coloc
