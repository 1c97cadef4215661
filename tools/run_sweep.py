"""Back-to-back 10k x 10k K2NN sweeps through the C ABI (for rocprofv3 passes).  usage: run_sweep.py [reps] [formulation]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ctx = Context(device=0, width=640, height=480, maxkp=10000, detector=False)
if len(sys.argv) > 2:
    ctx.set_k2nn_formulation(sys.argv[2])
Q, T = synth.planted_descriptors(10000, 10000, seed=3000)
dq, dt = torch.from_numpy(Q).cuda(), torch.from_numpy(T).cuda()
dm = torch.empty(10000, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    ctx.match_2nn_dev(dq.data_ptr(), 10000, dt.data_ptr(), 10000, 40, dm.data_ptr())
ctx.sync()
print("accepted", int((dm >= 0).sum()))
