// tools/fivept_host.cpp -- the five-point solver of csrc/fivept.h compiled for the HOST: hit rate / solution count on the
// same synthetic scenes as tools/fivept_bench.hip, for trying algorithm changes without a GPU.
// Build: g++ -O2 -std=c++17 -o /tmp/fivept_host tools/fivept_host.cpp
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "fivept_scene.h"
#include "../coloc_amd/csrc/fivept.h"

int main(int argc, char** argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 2000;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 7;
    std::vector<double> q1, q2, Etrue;
    fpt_make_scenes(S, q1, q2, Etrue, seed);
    std::vector<double> E(90 * (size_t)S, 0.0);
    std::vector<int> n(S);
    static FptWorkspace ws;
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < S; ++s) {
        double a[5][2], b[5][2];
        for (int p = 0; p < 5; ++p) for (int c = 0; c < 2; ++c) { a[p][c] = q1[10 * s + 2 * p + c]; b[p][c] = q2[10 * s + 2 * p + c]; }
        n[s] = fivept_solve(a, b, &E[90 * (size_t)s], ws);
        for (int i = 9 * n[s]; i < 90; ++i) E[90 * (size_t)s + i] = 0.0;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int hits; long nsol; uint64_t h;
    fpt_score(S, E, n, Etrue, &hits, &nsol, &h);
    printf("S=%d  %.1f us per solve (host)  hit-rate %.4f  solutions/sample %.2f  checksum %016llx\n", S, 1e6 * dt / S, (double)hits / S, (double)nsol / S,
           (unsigned long long)h);
    return 0;
}
