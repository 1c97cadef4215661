#!/usr/bin/env python3
"""Soak test of clc_two_view_acransac under the models 'F' and 'H' against the sequential oracle (kinds 2 / 3, fed with the device's
minimal models): random sizes (5 .. 6000), outlier rates, seeds, iteration budgets and thresholds on ONE long-lived context, models
alternating (the workspace, parity copies and pinned blocks are reused from solve to solve, and by the resection solves run in
between); model, inlier list (order included), NFA, threshold and iteration count must be identical, every time.
usage: soak_two_view_models.py [runs]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, synth, oracle_lib
import twoview_host as tvh
from coloc_amd import Context

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = Context(device=0, detector=False, matcher=False)
orc = oracle_lib.Oracle()
rng = np.random.default_rng(23)
bad = 0
t0 = time.time()
for it in range(runs):
    model = "FH"[it % 2]
    n = int(rng.choice([5, 8, 9, 40, 64, 65, 128, 129, 300, 700, 1024, 1025, 1500, 2048, 2049, 3000, 4097, 6000]))
    outl = float(rng.choice([0.0, 0.2, 0.5, 0.7]))
    max_it = int(rng.choice([1, 9, 10, 40, 256, 300]))
    prec = float(rng.choice([float("inf"), float("inf"), 4.0, 1.0]))
    seed = int(rng.integers(1, 1 << 30))
    sc = tvh.scene(n, int(rng.integers(1 << 20)), planar=model == "H", outlier_frac=outl)
    def fit(sample, sc=sc, model=model):
        mo = ctx.two_view_minimal(model, sc["x1"], sc["x2"], sc["wh"], np.array([sample], dtype=np.int32))[0]
        return [m for m in mo if not np.isnan(m).any()]
    got = ctx.two_view_acransac(model, sc["x1"], sc["x2"], sc["wh"], max_iteration=max_it, seed=seed, precision=prec)
    want = orc.acransac(2 if model == "F" else 3, sc["x1"], sc["x2"], np.eye(3), fit, max_iteration=max_it, seed=seed, precision=prec, img_wh=sc["wh"])
    ok = (got["M"] is not None) == want["found"] and got["iterations"] == want["iterations"] \
        and ((math.isinf(got["min_nfa"]) and math.isinf(want["min_nfa"])) or abs(got["min_nfa"] - want["min_nfa"]) <= 1e-12 * abs(want["min_nfa"])) \
        and np.array_equal(got["inliers"], want["inliers"].astype(np.int32)) \
        and (not want["found"] or (np.array_equal(got["M"].reshape(-1), want["model"]) and got["error_max"] == want["error_max"]))
    if not ok:
        bad += 1
        print("MISMATCH run %d: model %s n %d outliers %.1f max_it %d precision %s seed %d: found %s/%s iterations %d/%d inliers %d/%d"
              % (it, model, n, outl, max_it, prec, seed, got["M"] is not None, want["found"], got["iterations"], want["iterations"],
                 len(got["inliers"]), len(want["inliers"])), flush=True)
    if it % 7 == 3:      # a resection solve in between: the same device workspace under another kind
        ps = synth.pnp_scene(int(rng.choice([50, 500, 2000])), seed=int(rng.integers(1 << 20)))
        ctx.pnp_acransac(ps["X"], ps["x"], ps["K"], max_iteration=64, seed=seed)
    if it % 50 == 49:
        print("run %d, %d mismatches, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
print("soak_two_view_models: %d runs, %d mismatches" % (runs, bad))
sys.exit(1 if bad else 0)
