#!/usr/bin/env python3
"""Soak of the sweep's unequal-share plans (round 4): random shapes in the range where a single pair runs as one round (6 000..16 000 queries,
2 500..30 000 train rows), the three matrix/popcount formulations in turn, every result (indices and both distances) compared with the CPU
oracle; reports how many shapes took each plan.  Not part of the test suite (minutes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
from oracle_lib import Oracle
from coloc_amd import Context

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    orc = Oracle()
    ctx = Context(device=0, width=640, height=480, maxkp=210001, detector=False)
    rng = np.random.default_rng(2024)
    took = {"per-XCD": 0, "all ids": 0, "equal": 0}
    t0 = time.time()
    for it in range(n):
        nq = int(rng.integers(6000, 16001))
        if it % 3 == 0:
            nq = int(rng.integers(24, 64)) * 256 - int(rng.integers(0, 256))     # a whole number of query blocks (often a multiple of 8)
        nt = int(rng.integers(2500, 30001))
        if it % 10 == 9:
            nt = int(rng.integers(135000, 210001))                                # beyond 4096 train tiles: the per-XCD table's begin field
        Q, T = synth.planted_descriptors(nq, nt, seed=int(rng.integers(1 << 30)))
        k = int(rng.integers(0, nt)); T[rng.integers(0, nt, 40)] = T[k]; Q[0] = T[k]     # ties for the minimum spread over the splits
        thr = int(rng.integers(0, 120))
        if nt > 40000:                                                            # the OpenMP loop for the big ones (indices only)
            mo, bo, so = orc.k2nn_omp(Q, T, rule=0, threshold=thr, kernel=0)[0], None, None
        else:
            mo, bo, so = orc.k2nn(Q, T, thr, want_dist=True)
        form = ("matrix", "popcount")[it & 1]
        ctx.set_k2nn_formulation(form)
        p = ctx.k2nn_plan_query(nq, nt)
        took["equal" if not p["bias_a_tiles"] else ("per-XCD" if p["qblocks"] % 8 == 0 else "all ids")] += 1
        m, b, s = ctx.match_2nn(Q, T, thr, want_dist=True)
        assert np.array_equal(m, mo) and (bo is None or (np.array_equal(b, bo) and np.array_equal(s, so))), (it, nq, nt, thr, form, p)
        if it % 20 == 0:
            print("%d shapes ok (%.0f s) plans so far %s" % (it + 1, time.time() - t0, took), flush=True)
    print("soak ok: %d shapes, plans %s" % (n, took))
    ctx.close()

if __name__ == "__main__":
    main()
