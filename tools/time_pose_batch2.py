#!/usr/bin/env python3
"""Batch size sweep of clc_pnp_localize_ac_batch (N = 1300 correspondences, 5 % outliers -- the streaming loop's map matches), refine on."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, synth
from coloc_amd import Context
from coloc_amd.abi import pnp_localize_batch
ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(8)]
N = int(os.environ.get("N", "1300")); OUT = float(os.environ.get("OUTL", "0.05"))
scenes = [synth.pnp_scene(N, seed=4000 + c, outlier_frac=OUT) for c in range(8)]
probs = [(s["X"], s["x"], s["K"]) for s in scenes]
for ncam in (1, 2, 3, 4, 6, 8):
    for _ in range(5): pnp_localize_batch(ctxs[:ncam], probs[:ncam], max_iteration=256, seeds=list(range(11, 11 + ncam)), refine=True)
    tb = []
    for rep in range(100):
        t0 = time.perf_counter(); got = pnp_localize_batch(ctxs[:ncam], probs[:ncam], max_iteration=256, seeds=list(range(11, 11 + ncam)), refine=True); tb.append(time.perf_counter() - t0)
    print("GPU_MAX_HW_QUEUES=%s  %d solves in a batch: p50 %.3f ms = %.3f per pose   (iterations %s)" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), ncam, np.median(tb) * 1e3, np.median(tb) * 1e3 / ncam, [g["iterations"] for g in got]), flush=True)
