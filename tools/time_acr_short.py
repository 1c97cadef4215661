"""p50 of the a-contrario pose solve at N = 1000 (30 % outliers), with and without refinement (run on the GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from coloc_amd import Context

ctx = Context(device=0, width=640, height=480, maxkp=10000)
for n in (int(a) for a in (sys.argv[1:] or ["1000"])):
    sc = synth.pnp_scene(n, seed=4000 + n, outlier_frac=0.3)
    for refine in (False, True):
        for rep in range(3):
            ts, rounds = [], []
            for it in range(100):
                t0 = time.perf_counter()
                r = ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], seed=it + 1, refine=refine)
                ts.append((time.perf_counter() - t0) * 1e3)
                rounds.append(r.get("rounds", 0))
            ts = np.sort(ts[10:])
            print("N=%d refine=%d: p50 %.3f ms  p10 %.3f  p95 %.3f  rounds median %s" % (n, refine, ts[len(ts) // 2], ts[len(ts) // 10], ts[int(len(ts) * .95)], np.median(rounds)))
