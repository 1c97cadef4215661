// tools/fivept_scene.h -- synthetic two-view scenes + hit-rate scoring shared by fivept_bench.hip and fivept_host.cpp
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

inline void fpt_make_scenes(int S, std::vector<double>& q1, std::vector<double>& q2, std::vector<double>& Etrue, uint64_t seed = 7)
{
    q1.assign(10 * S, 0.0); q2.assign(10 * S, 0.0); Etrue.assign(9 * S, 0.0);
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int s = 0; s < S; ++s) {
        // random rotation (small-ish) + translation, 5 points in front of both cameras
        double ax[3] = { 0.3 * U(rng), 0.3 * U(rng), 0.3 * U(rng) }, t[3] = { U(rng), U(rng), 0.3 * U(rng) };
        const double th = std::sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + 1e-12;
        const double k[3] = { ax[0] / th, ax[1] / th, ax[2] / th }, c = std::cos(th), sn = std::sin(th);
        double R[9] = { c + k[0] * k[0] * (1 - c), k[0] * k[1] * (1 - c) - k[2] * sn, k[0] * k[2] * (1 - c) + k[1] * sn,
                        k[1] * k[0] * (1 - c) + k[2] * sn, c + k[1] * k[1] * (1 - c), k[1] * k[2] * (1 - c) - k[0] * sn,
                        k[2] * k[0] * (1 - c) - k[1] * sn, k[2] * k[1] * (1 - c) + k[0] * sn, c + k[2] * k[2] * (1 - c) };
        const double tx[9] = { 0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0 };
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Etrue[9 * s + 3 * r + cc] = tx[3 * r] * R[cc] + tx[3 * r + 1] * R[3 + cc] + tx[3 * r + 2] * R[6 + cc];
        for (int p = 0; p < 5; ++p) {
            const double X[3] = { 2.0 * U(rng), 2.0 * U(rng), 4.0 + 2.0 * U(rng) };
            double Y[3];
            for (int r = 0; r < 3; ++r) Y[r] = R[3 * r] * X[0] + R[3 * r + 1] * X[1] + R[3 * r + 2] * X[2] + t[r];
            q1[10 * s + 2 * p] = X[0] / X[2]; q1[10 * s + 2 * p + 1] = X[1] / X[2];
            q2[10 * s + 2 * p] = Y[0] / Y[2]; q2[10 * s + 2 * p + 1] = Y[1] / Y[2];
        }
    }
}

// hits = samples whose solution set contains the true E (up to scale and sign, 1e-6 on the unit-norm entries)
inline void fpt_score(int S, const std::vector<double>& E, const std::vector<int>& n, const std::vector<double>& Etrue, int* hits_out, long* nsol_out,
                      uint64_t* checksum)
{
    int hits = 0; long nsol = 0; uint64_t h = 1469598103934665603ull;
    for (int s = 0; s < S; ++s) {
        nsol += n[s];
        double best = 1e30, nt = 0;
        for (int i = 0; i < 9; ++i) nt += Etrue[9 * s + i] * Etrue[9 * s + i];
        nt = std::sqrt(nt);
        for (int k = 0; k < n[s]; ++k) {
            const double* e = &E[90 * s + 9 * k];
            double ne = 0; for (int i = 0; i < 9; ++i) ne += e[i] * e[i]; ne = std::sqrt(ne);
            double dp = 0, dm = 0;
            for (int i = 0; i < 9; ++i) { const double a = e[i] / ne, b = Etrue[9 * s + i] / nt; dp = std::fmax(dp, std::fabs(a - b)); dm = std::fmax(dm, std::fabs(a + b)); }
            best = std::fmin(best, std::fmin(dp, dm));
        }
        hits += best < 1e-6;
        for (int i = 0; i < 90; ++i) { uint64_t b; double v = E[90 * s + i]; memcpy(&b, &v, 8); h = (h ^ b) * 1099511628211ull; }
    }
    *hits_out = hits; *nsol_out = nsol; *checksum = h;
}
