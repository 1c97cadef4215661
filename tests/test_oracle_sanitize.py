"""AddressSanitizer + UBSan over the CPU oracle (GPU sanitizers are not available on the pool): a C
driver exercises every oracle entry point on random / edge-shaped inputs."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "clc_oracle.h"
static unsigned long long st = 88172645463325252ull;
static unsigned rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (unsigned)(st >> 16); }
int main(void) {
    enum { NQ = 300, NT = 257, W = 101, H = 67 };
    uint8_t* q = malloc(NQ * 64), *t = malloc(NT * 64);
    for (int i = 0; i < NQ * 64; ++i) q[i] = (uint8_t)rnd();
    for (int i = 0; i < NT * 64; ++i) t[i] = (uint8_t)rnd();
    int32_t* m = malloc(NQ * 4); uint16_t* b = malloc(NQ * 2), *s = malloc(NQ * 2);
    orc_k2nn(q, NQ, t, NT, 40, m, b, s); orc_k2nn(q, NQ, t, 0, 40, m, NULL, NULL); orc_k2nn(q, 0, t, NT, 40, m, b, s);
    orc_k2nn_split(q, NQ, t, NT, 40, 7, m); orc_k2nn_split(q, NQ, t, 1, 40, 64, m);
    orc_k2nn_omp(q, NQ, t, NT, 0, 40, 0.8f, m); orc_k2nn_omp(q, NQ, t, NT, 1, 40, 0.8f, m);
    uint32_t w[8], h[8]; float f[8];
    orc_pyramid_dims(W, H, 1.2f, 8, w, h, f);
    uint8_t* lv[8]; size_t pitch[8];
    lv[0] = malloc((size_t)W * H); pitch[0] = W;
    for (int i = 0; i < W * H; ++i) lv[0][i] = (uint8_t)rnd();
    for (int i = 1; i < 8; ++i) { lv[i] = malloc((size_t)w[i] * h[i]); pitch[i] = w[i]; orc_lerp(lv[0], W, H, W, f[i], f[i], lv[i], w[i], h[i], w[i]); }
    enum { NK = 200 };
    orc_keypoint* kp = malloc(NK * sizeof *kp);
    for (int i = 0; i < NK; ++i) { kp[i].scale = (uint8_t)(rnd() % 8); kp[i].x = (int)(rnd() % w[kp[i].scale]); kp[i].y = (int)(rnd() % h[kp[i].scale]);
        kp[i].angle = ((int)(rnd() % 62832) - 31416) * 1e-4f; kp[i].score = 0; }
    uint8_t* d = malloc(NK * 64);
    orc_clatch((const uint8_t* const*)lv, w, h, pitch, kp, NK, d);
    orc_keypoint* fk = malloc(5000 * sizeof *fk);
    /* FAST reads rows only inside the image; the restatement must not copy the reference's over-reads */
    int n = orc_fast9(lv[0], W, H, W, 20, fk, 5000);
    for (int i = 0; i < n; ++i) (void)orc_feature_angle(lv[0], fk[i].x, fk[i].y, W);
    n = orc_fast9(lv[7], (int)w[7], (int)h[7], (int)w[7], 5, fk, 10);
    float* feat = malloc(NK * 16); orc_features_from_kps(kp, NK, feat);
    double Rt[24] = {1,0,0,0, 0,1,0,0, 0,0,1,5,  1,0,0,.1, 0,1,0,.1, 0,0,1,6}, K[9] = {500,0,50, 0,500,30, 0,0,1};
    double X[30], x[20], e[20]; for (int i = 0; i < 30; ++i) X[i] = (rnd() % 100) * 0.01; for (int i = 0; i < 20; ++i) x[i] = rnd() % 100;
    orc_pnp_residuals(Rt, 2, X, x, 10, K, e); int32_t c[2]; double cost[2]; orc_pnp_score(e, 2, 10, 16.0, c, cost);
    printf("ok %d %d\n", m[0], n);
    return 0;
}
'''


def test_oracle_under_asan_ubsan(tmp_path):
    drv = tmp_path / "drv.c"
    drv.write_text(DRIVER)
    exe = tmp_path / "drv"
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-ffp-contract=off", "-fopenmp", "-I", os.path.join(ROOT, "oracle"), str(drv),
                           os.path.join(ROOT, "oracle", "clc_oracle.c"), "-o", str(exe), "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="2")
    out = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("ok")
