"""Where the host side of one a-contrario pose solve goes: the Python wrapper against the bare C call (run on the GPU box)."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from coloc_amd import Context
from coloc_amd.abi import _p

ctx = Context(device=0, width=640, height=480, maxkp=10000)
n = 1000
sc = synth.pnp_scene(n, seed=4000 + n, outlier_frac=0.3)
X = np.ascontiguousarray(sc["X"], dtype=np.float64); x = np.ascontiguousarray(sc["x"], dtype=np.float64); K = np.ascontiguousarray(sc["K"], dtype=np.float64).reshape(9)
Rt = np.zeros(12); mask = np.zeros(n, dtype=np.uint8); inl = np.zeros(n, dtype=np.int32); cov = np.zeros(36)
ni, its = C.c_int(), C.c_int(); emax, nfa, rmse = C.c_double(), C.c_double(), C.c_double()
pX, px, pK, pRt, pm, pi, pc = _p(X), _p(x), _p(K), _p(Rt), _p(mask), _p(inl), _p(cov)
for name in ("wrapper", "bare", "bare+refine", "wrapper+refine"):
    ts = []
    for it in range(110):
        t0 = time.perf_counter()
        if name == "wrapper": ctx.pnp_acransac(X, x, K, seed=it + 1)
        elif name == "wrapper+refine": ctx.pnp_acransac(X, x, K, seed=it + 1, refine=True)
        elif name == "bare":
            ctx.lib.clc_pnp_acransac(ctx.h, pX, px, n, pK, 256, it + 1, float("inf"), pRt, pm, pi, C.byref(ni), C.byref(emax), C.byref(nfa), C.byref(its))
        else:
            ctx.lib.clc_pnp_localize_ac(ctx.h, pX, px, n, pK, 256, it + 1, float("inf"), 16.0, pRt, pc, pm, pi, C.byref(ni), C.byref(emax), C.byref(rmse))
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.sort(ts[10:])
    print("%-15s p50 %.3f ms  p10 %.3f  p95 %.3f" % (name, ts[len(ts) // 2], ts[len(ts) // 10], ts[int(len(ts) * .95)]))
