#!/usr/bin/env python3
"""Soak test of clc_pnp_localize (pinned staging read/written by the kernels themselves): 600 problems of random size on
a long-lived context against a fresh context each, results must be identical; poses are checked against ground truth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, synth
from coloc_amd import Context
a = Context(device=0, width=640, height=480, maxkp=1024)
rng = np.random.default_rng(5)
bad = 0
for it in range(600):
    N = int(rng.integers(3, 8000)); S = int(rng.choice([1, 7, 64, 256, 300]))
    sc = synth.pnp_scene(N, seed=int(rng.integers(1 << 20)))
    b = Context(device=0, width=640, height=480, maxkp=1024)          # fresh buffers every time
    ra = a.pnp_localize(sc["X"], sc["x"], sc["K"], n_samples=S, seed=it + 1, thr2=16.0)
    rb = b.pnp_localize(sc["X"], sc["x"], sc["K"], n_samples=S, seed=it + 1, thr2=16.0)
    b.close()
    for u, v in zip(ra, rb):
        if (u is None) != (v is None) or (u is not None and not np.array_equal(np.asarray(u), np.asarray(v))):
            bad += 1; print("mismatch", it, N, S); break
    if ra[0] is not None and N >= 50 and S >= 64:
        R, t = ra[0][:, :3], ra[0][:, 3]
        err = np.abs(R - sc["R"]).max() + np.abs(t - sc["t"]).max()
        if err > 0.05: print("pose far from truth", it, N, S, err)
    if it % 100 == 0: print(it, "ok", flush=True)
print("soak done, mismatches", bad)
