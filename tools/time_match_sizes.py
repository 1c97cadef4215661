#!/usr/bin/env python3
"""Device time of clc_match_2nn_dev over a list of (nq, nt) for the library COLOC_HIP_LIB points to; results checked against
the popcount formulation of the same library.  usage: time_match_sizes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
dev = torch.device("cuda", 0)
ctx = Context(device=0, width=640, height=480, maxkp=20000)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
print(os.path.basename(os.environ.get("COLOC_HIP_LIB", "in tree")))
SIZES = [(10000, 10000), (8508, 9390), (9000, 9000), (7000, 10000), (5000, 10000), (3000, 10000), (1500, 10000), (12000, 12000), (6000, 6000)]
if os.environ.get("SIZES"): SIZES = [tuple(int(v) for v in p.split("x")) for p in os.environ["SIZES"].split(",")]
for nq, nt in SIZES:
    Qh, Th = synth.planted_descriptors(nq, nt, seed=5)
    Q, T = torch.from_numpy(Qh).to(dev), torch.from_numpy(Th).to(dev)
    m = torch.empty(nq, dtype=torch.int32, device=dev); ref = torch.empty(nq, dtype=torch.int32, device=dev)
    ctx.set_k2nn_formulation("matrix")
    for _ in range(300): ctx.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(600): ctx.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    bb = (time.perf_counter() - t0) / 600 * 1e6
    ctx.set_k2nn_formulation("popcount")
    ctx.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, ref.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    print("  %6d x %6d: %7.2f us/launch back to back  (%.2f ns per 1000 comparisons)  %s" % (nq, nt, bb, bb * 1e3 / (nq * nt / 1e3), "identical" if torch.equal(m, ref) else "DIFFERENT"))
ctx.close()
