#!/usr/bin/env python3
"""LM iterations and wall time of clc_pnp_refine started from the RANSAC pose (bench scenes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
from coloc_amd import Context

ctx = Context(device=0, width=640, height=480, maxkp=1024)
for N in (200, 1000, 5000):
    sc = synth.pnp_scene(N, seed=4000 + N)
    its, ts = [], []
    for rep in range(30):
        Rt, mask, _ = ctx.pnp_ransac(sc["X"], sc["x"], sc["K"], n_samples=256, seed=rep + 1, thr2=16.0)
        t0 = time.perf_counter()
        R2, cov, rmse, it = ctx.pnp_refine(sc["X"], sc["x"], sc["K"], Rt, mask=mask.astype(np.uint8), huber_a=16.0)
        ts.append(time.perf_counter() - t0); its.append(it)
    print("N=%d  iterations min/median/max %d/%d/%d   refine call p50 %.1f us" % (N, min(its), sorted(its)[15], max(its), sorted(ts)[15] * 1e6))
ctx.close()
