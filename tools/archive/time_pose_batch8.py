#!/usr/bin/env python3
"""clc_pnp_localize_ac_batch with 8 poses, 30 calls (for rocprofv3 --kernel-trace: the timeline of the batch's launches; CLC_ACR_LOCKSTEP=0|1)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, synth
from coloc_amd import Context
from coloc_amd.abi import pnp_localize_batch
ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(8)]
scenes = [synth.pnp_scene(1000, seed=4000 + c, outlier_frac=0.3) for c in range(8)]
probs = [(s["X"], s["x"], s["K"]) for s in scenes]
tb = []
for rep in range(30):
    t0 = time.perf_counter(); pnp_localize_batch(ctxs, probs, max_iteration=256, seeds=list(range(11, 19)), refine=True); tb.append(time.perf_counter() - t0)
print("8 poses per call: p50 %.3f ms" % (np.median(tb[5:]) * 1e3))
