#!/usr/bin/env python3
"""Device time of clc_match_2nn_dev at 10k x 10k for the library COLOC_HIP_LIB points to (ablation builds: results are not
checked).  500 untimed launches first (clock settling), then median / min of 200 event-bracketed launches and a back-to-back figure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
dev = torch.device("cuda", 0)
ctx = Context(device=0, width=640, height=480, maxkp=20000)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
nq = nt = 10000
Qh, Th = synth.planted_descriptors(nq, nt, seed=5)
Q, T = torch.from_numpy(Qh).to(dev), torch.from_numpy(Th).to(dev)
m = torch.empty(nq, dtype=torch.int32, device=dev)
for _ in range(500): ctx.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
ts = []
for rep in range(200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); ctx.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream); b.record(st)
    b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
t0 = time.perf_counter()
for _ in range(1000): ctx.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
bb = (time.perf_counter() - t0) / 1000 * 1e6
print("%-28s events median %6.2f us  min %6.2f   back-to-back %6.2f us/launch" % (os.path.basename(os.environ.get("COLOC_HIP_LIB", "in tree")), sorted(ts)[100], min(ts), bb))
ctx.close()
