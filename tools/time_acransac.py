"""p50 of the a-contrario pose solve / two-view filter on host buffers (run on the GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from coloc_amd import Context
from test_gpu_epipolar import _two_view

ctx = Context(device=0, width=640, height=480, maxkp=10000)
for n in (200, 1000, 5000):
    for outl in (0.3, 0.6):
        sc = synth.pnp_scene(n, seed=4000 + n, outlier_frac=outl)
        for refine in (False, True):
            ts, its = [], []
            for it in range(60):
                t0 = time.perf_counter()
                r = ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], seed=it + 1, refine=refine)
                ts.append((time.perf_counter() - t0) * 1e3)
                its.append(r.get("iterations", 0))
            ts = np.sort(ts[5:])
            print("pnp N=%d outl=%.1f refine=%d: p50 %.3f ms p95 %.3f ms, iterations median %s, inliers %d" %
                  (n, outl, refine, ts[len(ts) // 2], ts[int(len(ts) * .95)], np.median(its), len(r["inliers"])))
old = []
sc = synth.pnp_scene(1000, seed=5000)
for it in range(60):
    t0 = time.perf_counter(); ctx.pnp_localize(sc["X"], sc["x"], sc["K"], n_samples=256, seed=it + 1, thr2=16.0); old.append((time.perf_counter() - t0) * 1e3)
print("fixed-threshold localize N=1000: p50 %.3f ms" % np.median(old[5:]))
x1, x2, F, out = _two_view(1000, seed=22)
K = synth.pnp_scene(5, seed=22)["K"]
ts = []
for it in range(40):
    t0 = time.perf_counter(); r = ctx.essential_acransac(x1, x2, K, K, (1280, 720), seed=it + 1); ts.append((time.perf_counter() - t0) * 1e3)
print("essential acransac N=1000: p50 %.3f ms, iterations %d, inliers %d" % (np.median(ts[5:]), r["iterations"], len(r["inliers"])))
ts = []
for it in range(40):
    t0 = time.perf_counter(); ctx.essential_ransac(x1, x2, K, K, n_samples=256, seed=it + 1, thr2=4.0); ts.append((time.perf_counter() - t0) * 1e3)
print("essential fixed-threshold N=1000: p50 %.3f ms" % np.median(ts[5:]))
