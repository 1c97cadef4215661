"""A few a-contrario pose solves with refinement (for rocprofv3: tools/trace_timeline.py).  usage: trace_acr_refine.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from coloc_amd import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = Context(device=0, width=640, height=480, maxkp=10000)
sc = synth.pnp_scene(n, seed=4000 + n, outlier_frac=0.3)
for it in range(12):
    ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], seed=it + 1, refine=True)
