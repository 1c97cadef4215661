#!/usr/bin/env python3
"""Soak of clc_essential_acransac (round 5: two launches per round on parity-indexed copies, the solve shared with fivept_kernel) against the
sequential oracle: random sizes (6 .. 3000), outlier rates, seeds, iteration budgets, on ONE long-lived context and through the batch
entry; model (F and E), inlier list (order included), NFA, threshold and iteration count must be identical, every time.
usage: soak_essential.py [runs]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, synth, oracle_lib
from coloc_amd import Context
from coloc_amd.abi import essential_acransac_batch
import test_gpu_two_view_batch as tv
from test_gpu_acransac import _f_from_e      # F = K2^-T E K1^-1 in the device's operation order (the oracle's residuals must see the device's bits)

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
orc = oracle_lib.Oracle()
ctx = Context(device=0, detector=False, matcher=False)
pool = [Context(device=0, detector=False, matcher=False) for _ in range(3)]
rng = np.random.default_rng(99)
K, WH = tv.K, tv.WH
t0 = time.time()
found = 0
for it in range(runs):
    n = int(rng.integers(6, 3001)) if it % 7 else int(rng.choice([6, 7, 12, 13, 64, 65, 1024, 1025, 2048, 2049]))
    p = tv._pair(1000 + it, n=n, outliers=float(rng.uniform(0.0, 0.6)), noise=float(rng.uniform(0.1, 1.0)))
    seed = int(rng.integers(1, 1 << 30)); max_it = int(rng.choice([8, 32, 64, 256, 400]))
    x1, x2 = p["x1"], p["x2"]
    def fit(sample):
        Es = ctx.essential_fivepoint(x1, x2, K, K, np.array([sample], dtype=np.int32))[0]
        return [np.concatenate([_f_from_e(E, K, K), E]) for E in Es if not np.isnan(E).any()]
    want = orc.acransac(1, x1, x2, K, fit, max_iteration=max_it, seed=seed, img_wh=WH)
    got = ctx.essential_acransac(x1, x2, K, K, WH, max_iteration=max_it, seed=seed)
    gotb = essential_acransac_batch(pool[:1 + it % 3], [(x1, x2, K, K, WH, seed)] * (1 + it % 3), max_iteration=max_it)
    for g in [got] + gotb:
        assert g["iterations"] == want["iterations"], (it, n, seed, max_it, g["iterations"], want["iterations"])
        if not want["found"]:
            assert g["E"] is None, (it, n)
            continue
        assert abs(g["min_nfa"] - want["min_nfa"]) <= 1e-12 * abs(want["min_nfa"]) and g["error_max"] == want["error_max"], (it, n, seed)
        assert np.array_equal(g["inliers"], want["inliers"].astype(np.int32)), (it, n, seed)
        assert np.array_equal(g["F"].reshape(-1), want["model"][:9]) and np.array_equal(g["E"].reshape(-1), want["model"][9:]), (it, n, seed)
    found += bool(want["found"])
    if it % 20 == 0:
        print("%d problems ok (%.0f s), %d with a model" % (it + 1, time.time() - t0, found), flush=True)
print("soak ok: %d two-view problems, %d with a meaningful model" % (runs, found))
