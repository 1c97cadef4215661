#!/usr/bin/env python3
"""A/B of the K2NN sweep with equal shares against unequal shares by wave slot, alternated in ONE process on one device (boxes differ by more
than the effect): 10k x 10k by default (SIZES=nqxnt,...), blocks of 300 back-to-back launches, 12 alternations, median per setting."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
def make(bias):
    old = os.environ.get("CLC_K2NN_BIAS")
    os.environ["CLC_K2NN_BIAS"] = bias
    try:
        return Context(device=0, width=640, height=480, maxkp=70000, detector=False)
    finally:
        if old is None: del os.environ["CLC_K2NN_BIAS"]
        else: os.environ["CLC_K2NN_BIAS"] = old
settings = os.environ.get("BIASES", "0,0;295,264").split(";")
ctxs = [make(b) for b in settings]
sizes = [tuple(int(v) for v in p.split("x")) for p in os.environ.get("SIZES", "10000x10000").split(",")]
for nq, nt in sizes:
    Qh, Th = synth.planted_descriptors(nq, nt, seed=5)
    Q, T = torch.from_numpy(Qh).to(dev), torch.from_numpy(Th).to(dev)
    ms = [torch.empty(nq, dtype=torch.int32, device=dev) for _ in ctxs]
    for c, m in zip(ctxs, ms):
        for _ in range(300): c.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    t = [[] for _ in ctxs]
    for rep in range(12):
        for k, (c, m) in enumerate(zip(ctxs, ms)):
            t0 = time.perf_counter()
            for _ in range(300): c.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream)
            torch.cuda.synchronize()
            t[k].append((time.perf_counter() - t0) / 300 * 1e6)
    same = all(torch.equal(ms[0], m) for m in ms[1:])
    print("%6d x %6d: " % (nq, nt) + "   ".join("bias %-8s %.2f us (min %.2f) plan %s" % (b, np.median(v), min(v), (lambda p: (p["splits"], p["bias_a_tiles"], p["bias_b_tiles"]))(c.k2nn_plan_query(nq, nt)))
                                               for b, v, c in zip(settings, t, ctxs)) + ("   identical" if same else "   DIFFERENT"), flush=True)
for c in ctxs: c.close()
