// ci_driver.cpp -- reads "CA(9) CB(9) ca(3) cb(3)" per line from stdin, prints minX minValue covFused(9) poseFused(3).
#include <cstdio>
#include "HIPCovIntersection.hpp"
int main()
{
    coloc::Mat3d A, B; coloc::Vec3d a, b;
    for (;;) {
        for (int i = 0; i < 9; ++i) if (std::scanf("%lf", &A[i]) != 1) return 0;
        for (int i = 0; i < 9; ++i) if (std::scanf("%lf", &B[i]) != 1) return 0;
        for (int i = 0; i < 3; ++i) if (std::scanf("%lf", &a[i]) != 1) return 0;
        for (int i = 0; i < 3; ++i) if (std::scanf("%lf", &b[i]) != 1) return 0;
        coloc::HIPCovIntersection ci;
        ci.loadData(A, B, a, b);
        ci.optimize();
        ci.computeFusedValues();
        std::printf("%.17g %.17g", ci.minX, ci.minValue);
        for (double v : ci.covFused) std::printf(" %.17g", v);
        for (double v : ci.poseFused) std::printf(" %.17g", v);
        std::printf("\n");
    }
}
