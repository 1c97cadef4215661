#!/usr/bin/env python3
"""Experiment: one 10k x 10k pair as TWO jobs of one launch (early query blocks / late query blocks) with their own split counts
(CLC_K2NN_SPLITS_PER_JOB), so that the workgroups dispatched third onto their CU get less work.  usage: k2nn_bias_lab.py qsplit sA sB
NOTE: the planner hook that read CLC_K2NN_SPLITS_PER_JOB was an experiment and is NOT in the tree (result: profiles/r03_k2nn_pipelined_pmc.txt);
without it both jobs get the planner's own split count and this script only times the two-job form of the pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, synth
qsplit, sA, sB = (int(a) for a in sys.argv[1:4])
if sA > 0: os.environ["CLC_K2NN_SPLITS_PER_JOB"] = "%d,%d" % (sA, sB)
from coloc_amd import Context
dev = torch.device("cuda", 0)
ctx = Context(device=0, width=640, height=480, maxkp=20000)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
nq = nt = 10000
Qh, Th = synth.planted_descriptors(nq, nt, seed=5)
arena = torch.from_numpy(np.concatenate([Qh, Th])).to(dev)
m = torch.empty(nq, dtype=torch.int32, device=dev)
ref = torch.empty(nq, dtype=torch.int32, device=dev)
jobs = [(0, nq, nq, nt, 0, 40)] if qsplit <= 0 else [(0, qsplit, nq, nt, 0, 40), (qsplit, nq - qsplit, nq, nt, qsplit, 40)]
run = lambda out: ctx.match_jobs_dev(arena.data_ptr(), jobs, out.data_ptr(), st.cuda_stream)
for _ in range(500): run(m)
torch.cuda.synchronize()
ts = []
for rep in range(200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); run(m); b.record(st); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
t0 = time.perf_counter()
for _ in range(1000): run(m)
torch.cuda.synchronize()
bb = (time.perf_counter() - t0) / 1000 * 1e6
os.environ.pop("CLC_K2NN_SPLITS_PER_JOB", None)
ctx.match_jobs_dev(arena.data_ptr(), [(0, nq, nq, nt, 0, 40)], ref.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
print("qsplit %5d splits %2d/%2d: events median %6.2f us  min %6.2f   back-to-back %6.2f us/launch   %s" % (qsplit, sA, sB, sorted(ts)[100], min(ts), bb, "identical" if torch.equal(m, ref) else "DIFFERENT"))
ctx.close()
