"""Phase stamps of acr_round_kernel (library built with CLC_EXTRA_FLAGS=-DCLC_ACR_STAMP): slot workgroup 0, thread 0, last launch that
evaluated a batch.  Run on the GPU box."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from coloc_amd import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = Context(device=0, width=640, height=480, maxkp=10000)
sc = synth.pnp_scene(n, seed=4000 + n, outlier_frac=0.3)
names = ["start", "replayed", "barrier", "sample ids", "p3p", "barrier", "residuals", "sorted", "exact keys", "nfa + end", "-", "(replay: loads landed)", "(replay: reductions done)", "(own NFA terms)", "(wave reduce)", "(barrier)"]
acc = []
for it in range(30):
    ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], seed=it + 1)
    st = (C.c_ulonglong * 16)()
    assert ctx.lib.clc_debug_acr_stamps(st) == 0
    acc.append([st[i] - st[0] for i in range(16)])
acc = np.median(np.array(acc[5:], dtype=np.float64), axis=0)
prev = 0.0
for nm, v in zip(names, acc):
    print("%-22s %8.0f cycles  (+%6.0f)" % (nm, v, v - prev)); prev = v
