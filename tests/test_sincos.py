"""The product's deterministic sin/cos (coloc_amd/csrc/clc_sincos.h, compiled here for the HOST with
g++ -ffp-contract=off) against libm evaluated in double and rounded once -- the oracle's definition."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
#include "clc_sincos.h"
int main() {
    long bad = 0, n = 0;
    for (uint32_t u = 0; u <= 0x40490fdbu; u += 211) {          // every 211th float in [0, pi], both signs
        for (int sgn = 0; sgn < 2; ++sgn) {
            uint32_t v = u | (sgn ? 0x80000000u : 0u); float a; memcpy(&a, &v, 4);
            float s, c; clc_sincosf(a, &s, &c);
            float s2 = (float)sin((double)a), c2 = (float)cos((double)a);
            ++n;
            if (s != s2 || c != c2) ++bad;                         // value compare: -0 == +0
        }
    }
    uint64_t st = 88172645463325252ull;
    for (int i = 0; i < 2000000; ++i) {                            // wider range: |a| < 1e4
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        float a = (float)(((double)(st >> 11) / 9007199254740992.0) * 2e4 - 1e4);
        float s, c; clc_sincosf(a, &s, &c);
        if (s != (float)sin((double)a) || c != (float)cos((double)a)) ++bad;
        ++n;
    }
    printf("%ld %ld\n", n, bad);
    return 0;
}
'''


def test_sincos_matches_correctly_rounded_libm(tmp_path):
    src = tmp_path / "sc.cpp"
    src.write_text(SRC)
    exe = tmp_path / "sc"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "coloc_amd", "csrc"),
                           str(src), "-o", str(exe)])
    n, bad = (int(v) for v in subprocess.check_output([str(exe)]).split())
    assert n > 10_000_000
    assert bad == 0
