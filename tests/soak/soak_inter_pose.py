#!/usr/bin/env python3
"""Soak of clc_inter_pose_batch against the numpy statement of the same chain (tests/test_gpu_two_view_batch.py _numpy_chain): random
worlds (100 .. 2000 correspondences, outlier rates, noise, map coverage), 1 .. 4 pairs per call: same front / common counts, scale to
1e-9, refined pose to 1e-6, or the same stage of failure.  usage: soak_inter_pose.py [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from coloc_amd import Context
from coloc_amd.abi import inter_pose_batch
import test_gpu_two_view_batch as tv
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(3)
ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(4)]
ok_n = fail_n = 0
t0 = time.time()
for it in range(calls):
    n = int(rng.integers(100, 2001))
    world = tv._pair(5000 + it, n=n, outliers=float(rng.uniform(0.0, 0.5)), noise=float(rng.uniform(0.1, 0.8)))
    if it % 6 == 5:                                                # thin the map: the scale stage may fail
        mi = world["map_index"].copy(); mi[rng.random(n) < 0.97] = -1; world["map_index"] = mi
    npair = 1 + it % 4
    probs = []
    for k in range(npair):
        q = dict(world); q["seed"] = int(rng.integers(1, 1 << 30)); probs.append(q)
    res = inter_pose_batch(ctxs[:npair], probs, world["map_X"])
    for q, r in zip(probs, res):
        assert r["status"] == 0
        if r["stage"] != 0:
            fail_n += 1
            assert r["stage"] in (1, 2, 3), r["stage"]
            continue
        ref = tv._numpy_chain(ctxs[0], q, r)
        assert r["n_front"] == ref["n_front"] and r["n_common"] == ref["n_common"], (it, r["n_front"], ref["n_front"], r["n_common"], ref["n_common"])
        assert abs(r["scale"] / ref["scale"] - 1) < 1e-9 and np.allclose(r["Rt"], ref["Rt"], atol=1e-6) and abs(r["rmse"] - ref["rmse"]) < 1e-6, (it, r["scale"], ref["scale"])
        ok_n += 1
    if it % 10 == 0:
        print("%d calls (%.0f s): %d pairs through, %d stopped at a stage" % (it + 1, time.time() - t0, ok_n, fail_n), flush=True)
print("soak ok: %d calls, %d pairs through, %d stopped at a stage" % (calls, ok_n, fail_n))
