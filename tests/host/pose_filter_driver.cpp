// pose_filter_driver.cpp -- drives HIPPoseFilter / HIPPoseLog from stdin for tests/test_pose_filter.py.
//   "F nDrones"                                      new filter
//   "M drone  t(3) R(9)"                             fillMeasurements
//   "U drone  rmse cov(36)"                          update -> prints R(9) t(3) gate rejected init P(36)
//   "E R(9)"                                         prints eulerAnglesZYX(3) and the remapped angles(3)
//   "L idx source dest R(9) c(3) cov(36) rmse n"     prints the CSV record
//   "P file nPoses nLandmarks xyz..."                logMaptoPLY(file) of those points, then logPosetoPLY(file + ".track") of every pose
#include <cstdio>
#include <iostream>
#include <memory>
#include "HIPPoseFilter.hpp"
#include "HIPPoseLog.hpp"
int main()
{
    std::unique_ptr<coloc::HIPPoseFilter> f;
    char cmd;
    while (std::scanf(" %c", &cmd) == 1) {
        if (cmd == 'F') { unsigned n; if (std::scanf("%u", &n) != 1) return 1; f.reset(new coloc::HIPPoseFilter(n)); }
        else if (cmd == 'M') {
            int d; std::array<double, 3> t; std::array<double, 9> R;
            if (std::scanf("%d", &d) != 1) return 1;
            for (auto& v : t) if (std::scanf("%lf", &v) != 1) return 1;
            for (auto& v : R) if (std::scanf("%lf", &v) != 1) return 1;
            f->fillMeasurements(f->droneMeasurements[(size_t)d], t, R);
        } else if (cmd == 'U') {
            int d; float rmse; coloc::Cov6d cov;
            if (std::scanf("%d %f", &d, &rmse) != 2) return 1;
            for (auto& v : cov) if (std::scanf("%lf", &v) != 1) return 1;
            std::array<double, 9> R; std::array<double, 3> t;
            f->update(d, cov, rmse, R, t);
            for (double v : R) std::printf("%.17g ", v);
            for (double v : t) std::printf("%.17g ", v);
            std::printf("%.17g %d %d", f->lastGateDistance, (int)f->lastRejected, (int)f->inInitialPhase());
            for (double v : f->droneFilters[(size_t)d].errorCovPost) std::printf(" %.17g", v);
            std::printf("\n");
        } else if (cmd == 'E') {
            std::array<double, 9> R;
            for (auto& v : R) if (std::scanf("%lf", &v) != 1) return 1;
            auto e = coloc::eulerAnglesZYX(R); auto g = e; coloc::convertAnglesForLogging(g);
            std::printf("%.17g %.17g %.17g %.17g %.17g %.17g\n", e[0], e[1], e[2], g[0], g[1], g[2]);
        } else if (cmd == 'L') {
            int idx, s, d, n; float rmse; std::array<double, 9> R; std::array<double, 3> c; std::array<double, 36> cov;
            if (std::scanf("%d %d %d", &idx, &s, &d) != 3) return 1;
            for (auto& v : R) if (std::scanf("%lf", &v) != 1) return 1;
            for (auto& v : c) if (std::scanf("%lf", &v) != 1) return 1;
            for (auto& v : cov) if (std::scanf("%lf", &v) != 1) return 1;
            if (std::scanf("%f %d", &rmse, &n) != 2) return 1;
            std::fflush(stdout);
            coloc::HIPPoseLog::writePoseCov(std::cout, idx, s, d, R, c, cov, rmse, n);
            std::cout.flush();
        } else if (cmd == 'P') {
            char name[512]; int np, nl;
            if (std::scanf("%511s %d %d", name, &np, &nl) != 3) return 1;
            std::vector<std::array<double, 3>> poses((size_t)np), pts((size_t)nl);
            for (auto& p : poses) for (auto& v : p) if (std::scanf("%lf", &v) != 1) return 1;
            for (auto& p : pts) for (auto& v : p) if (std::scanf("%lf", &v) != 1) return 1;
            coloc::HIPPoseLog log;
            bool ok = log.logMaptoPLY(poses, pts, std::string(name));
            for (const auto& p : poses) ok = log.logPosetoPLY(p, std::string(name) + ".track") && ok;
            std::printf("%d\n", (int)ok);
        } else return 2;
    }
    return 0;
}
