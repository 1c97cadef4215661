#!/usr/bin/env python3
"""clc_pnp_localize_ac_batch against the same solves one after the other (clc_pnp_localize_ac through Context.pnp_acransac):
BASELINE config[2]'s "batched PnP/RANSAC pose" -- 4 and 8 cameras, N = 200 / 1000 / 5000 correspondences each, 30 % outliers."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, synth
from coloc_amd import Context
from coloc_amd.abi import pnp_localize_batch
main = Context(device=0, detector=False, matcher=False)
for ncam in (4, 8):
    ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(ncam)]
    for N in (200, 1000, 5000):
        scenes = [synth.pnp_scene(N, seed=4000 + c, outlier_frac=0.3) for c in range(ncam)]
        probs = [(s["X"], s["x"], s["K"]) for s in scenes]
        seeds = [11 + c for c in range(ncam)]
        for refine in (False, True):
            for _ in range(5):
                pnp_localize_batch(ctxs, probs, max_iteration=256, seeds=seeds, refine=refine)
                for (X, x, K), sd in zip(probs, seeds): main.pnp_acransac(X, x, K, max_iteration=256, seed=sd, refine=refine)
            tb, ts = [], []
            for rep in range(60):
                t0 = time.perf_counter(); got = pnp_localize_batch(ctxs, probs, max_iteration=256, seeds=seeds, refine=refine); tb.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                want = [main.pnp_acransac(X, x, K, max_iteration=256, seed=sd, refine=refine) for (X, x, K), sd in zip(probs, seeds)]
                ts.append(time.perf_counter() - t0)
            same = all(np.array_equal(g["inliers"], w["inliers"]) and np.array_equal(g["Rt"], w["Rt"]) for g, w in zip(got, want))
            print("%d cameras x N=%5d refine=%d   batch p50 %.3f ms (%.3f per pose)   one after the other p50 %.3f ms (%.3f per pose)   identical=%s"
                  % (ncam, N, refine, np.median(tb) * 1e3, np.median(tb) * 1e3 / ncam, np.median(ts) * 1e3, np.median(ts) * 1e3 / ncam, same), flush=True)
    for c in ctxs: c.close()
main.close()
