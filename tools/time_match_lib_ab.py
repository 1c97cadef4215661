#!/usr/bin/env python3
"""A/B of two builds of the library in ONE process on one device (LIBS=a.so;b.so; each library is dlopen'ed under its own path), 10k x 10k sweep,
blocks of 300 back-to-back launches, 12 alternations, medians."""
import os, sys, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
import coloc_amd.abi as abi
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
libs = os.environ["LIBS"].split(";")
ctxs = []
for path in libs:
    os.environ["COLOC_HIP_LIB"] = path
    abi._LIB = None if hasattr(abi, "_LIB") else None
    for name in ("_lib", "_LIB", "_cached"):
        if hasattr(abi, name): setattr(abi, name, None)
    ctxs.append(abi.Context(device=0, width=640, height=480, maxkp=70000, detector=False))
for nq, nt in [tuple(int(v) for v in p.split("x")) for p in os.environ.get("SIZES", "10000x10000").split(",")]:
  Qh, Th = synth.planted_descriptors(nq, nt, seed=5)
  Q, T = torch.from_numpy(Qh).to(dev), torch.from_numpy(Th).to(dev)
  ms = [torch.empty(nq, dtype=torch.int32, device=dev) for _ in ctxs]
  for c, m in zip(ctxs, ms):
    for _ in range(300): c.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream)
  torch.cuda.synchronize()
  t = [[] for _ in ctxs]
  for rep in range(8):
    for k, (c, m) in enumerate(zip(ctxs, ms)):
        t0 = time.perf_counter()
        for _ in range(300): c.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize()
        t[k].append((time.perf_counter() - t0) / 300 * 1e6)
  print("%6d x %6d  " % (nq, nt) + "   ".join("%s %.2f us (min %.2f)" % (os.path.basename(p), np.median(v), min(v)) for p, v in zip(libs, t)), "  identical" if all(torch.equal(ms[0], m) for m in ms[1:]) else "  DIFFERENT", "  libs distinct:", len({id(c.lib) for c in ctxs}) == len(ctxs))
