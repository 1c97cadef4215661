#!/usr/bin/env python3
"""Soak test of the matcher's self re-arming workspace: many random shapes and repeated large calls on one context,
every result compared with the CPU oracle (tests/oracle_lib.py).  Not part of the test suite (minutes, not seconds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import synth
from oracle_lib import Oracle
from coloc_amd import Context

def main():
    n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    orc = Oracle()
    ctx = Context(device=0, width=640, height=480, maxkp=20000)
    rng = np.random.default_rng(99)
    t0 = time.time()
    for it in range(n_rand):
        nq, nt = int(rng.integers(1, 6000)), int(rng.integers(1, 6000))
        Q, T = synth.planted_descriptors(nq, nt, seed=int(rng.integers(1 << 30)))
        thr = int(rng.integers(0, 120))
        m = ctx.match_2nn(Q, T, thr)
        assert np.array_equal(m, orc.k2nn(Q, T, thr)), (it, nq, nt, thr)
        if it % 100 == 0:
            print("random shapes: %d ok (%.0f s)" % (it, time.time() - t0), flush=True)
    Q, T = synth.planted_descriptors(10000, 10000, seed=5)
    want = orc.k2nn(Q, T, 40)
    for it in range(n_big):
        assert np.array_equal(ctx.match_2nn(Q, T, 40), want), it
        if it % 50 == 0:
            print("10k x 10k repeats: %d ok (%.0f s)" % (it, time.time() - t0), flush=True)
    print("soak ok: %d random shapes, %d repeats of 10k x 10k" % (n_rand, n_big))
    ctx.close()

if __name__ == "__main__":
    main()
