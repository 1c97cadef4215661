#!/usr/bin/env python3
"""A/B of the K2NN formulations on the same descriptors: device time of clc_match_2nn_dev (HIP events, median of 60, interleaved
so that clock drift hits every variant alike) + a results check.  usage: time_match_ab.py [variants ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
variants = sys.argv[1:] or ["matrix", "matrix-plain"]
dev = torch.device("cuda", 0)
ctx = Context(device=0, width=640, height=480, maxkp=20000)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
for nq, nt in [(10000, 10000), (20000, 20000), (4000, 4000), (1000, 20000)]:
    Qh, Th = synth.planted_descriptors(nq, nt, seed=5)
    Q, T = torch.from_numpy(Qh).to(dev), torch.from_numpy(Th).to(dev)
    m = {v: torch.empty(nq, dtype=torch.int32, device=dev) for v in variants}
    ts = {v: [] for v in variants}
    for rep in range(65):
        for v in variants:
            ctx.set_k2nn_formulation(v)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); ctx.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m[v].data_ptr(), st.cuda_stream); b.record(st)
            b.synchronize()
            if rep >= 5: ts[v].append(a.elapsed_time(b) * 1e3)
    same = all(bool(torch.equal(m[variants[0]], m[v])) for v in variants)
    print("%6d x %6d  " % (nq, nt) + "  ".join("%s %7.2f us (min %7.2f)" % (v, sorted(ts[v])[len(ts[v]) // 2], min(ts[v])) for v in variants) + ("  identical" if same else "  DIFFERENT"))
ctx.close()
