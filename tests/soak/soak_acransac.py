#!/usr/bin/env python3
"""Soak test of clc_pnp_acransac / clc_pnp_localize_ac against the sequential oracle: random sizes (4 .. 6000), outlier rates, seeds,
iteration budgets and thresholds on ONE long-lived context (state, parity copies and pinned blocks are reused from solve to solve);
model, inlier list (order included), NFA, threshold and iteration count must be identical, every time.  usage: soak_acransac.py [runs]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, synth, oracle_lib
from coloc_amd import Context

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = Context(device=0, width=640, height=480, maxkp=1024)
orc = oracle_lib.Oracle()
rng = np.random.default_rng(17)
bad = 0
t0 = time.time()
for it in range(runs):
    n = int(rng.choice([4, 5, 9, 40, 64, 65, 128, 129, 300, 700, 1024, 1025, 1500, 2048, 2049, 3000, 4097, 6000]))
    outl = float(rng.choice([0.0, 0.2, 0.5, 0.7, 0.9]))
    max_it = int(rng.choice([1, 9, 10, 40, 256, 300]))
    prec = float(rng.choice([float("inf"), float("inf"), 4.0, 1.0]))
    seed = int(rng.integers(1, 1 << 30))
    sc = synth.pnp_scene(n, seed=int(rng.integers(1 << 20)), outlier_frac=outl)
    def fit(sample, sc=sc):
        h = ctx.pnp_p3p(sc["X"], sc["x"], sc["K"], np.array([sample], dtype=np.int32))[0]
        return [m for m in h if not np.isnan(m).any()]
    got = ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], max_iteration=max_it, seed=seed, precision=prec)
    want = orc.acransac(0, sc["X"], sc["x"], sc["K"], fit, max_iteration=max_it, seed=seed, precision=prec)
    ok = (got["Rt"] is not None) == want["found"] and got["iterations"] == want["iterations"] \
        and ((math.isinf(got["min_nfa"]) and math.isinf(want["min_nfa"])) or abs(got["min_nfa"] - want["min_nfa"]) <= 1e-12 * abs(want["min_nfa"])) \
        and np.array_equal(got["inliers"], want["inliers"].astype(np.int32))
    if ok and want["found"]:
        ok = np.array_equal(got["Rt"].reshape(-1), want["model"]) and got["error_max"] == want["error_max"]
    if ok and it % 3 == 0:          # the refining entry point must select the same set
        r2 = ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], max_iteration=max_it, seed=seed, precision=prec, refine=True)
        ok = np.array_equal(r2["inliers"], got["inliers"]) and (r2["Rt"] is None) == (got["Rt"] is None)
    if not ok:
        bad += 1
        print("MISMATCH run %d: n=%d outl=%.1f max_it=%d precision=%s seed=%d" % (it, n, outl, max_it, prec, seed), flush=True)
    if it % 50 == 0: print(it, "runs,", bad, "mismatches, %.0f s" % (time.time() - t0), flush=True)
print("soak done: %d runs, %d mismatches" % (runs, bad))
sys.exit(1 if bad else 0)
