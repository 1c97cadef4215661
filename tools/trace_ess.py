"""A few a-contrario two-view filters (for rocprofv3 --kernel-trace: tools/prof_any.sh).  usage: trace_ess.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from coloc_amd import Context
from test_gpu_epipolar import _two_view
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = Context(device=0, width=640, height=480, maxkp=10000)
x1, x2, F, out = _two_view(n, seed=22)
K = synth.pnp_scene(5, seed=22)["K"]
for it in range(12):
    r = ctx.essential_acransac(x1, x2, K, K, (1280, 720), seed=it + 1)
print(r["iterations"], r.get("rounds"), len(r["inliers"]))
