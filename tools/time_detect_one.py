"""detect kernels on one 640x480 (or WxH given) frame, 300 calls back to back: for rocprofv3 --kernel-trace --stats (per-kernel averages of
the library COLOC_HIP_LIB points to; ablation builds: results are not checked).  usage: time_detect_one.py [W H]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
ctx = Context(device=0, width=W, height=H, maxkp=20000)
img = torch.from_numpy(synth.rect_image(W, H, seed=1000, noise_sigma=2.0)).to(dev)
ctx.pyramid_build_dev(img.data_ptr(), W, H, W, s)
for _ in range(300):
    ctx.detect_dev(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    ctx.detect_dev(s)
torch.cuda.synchronize()
print("%s %dx%d: detect (two launches) %.2f us per call back to back" % (os.path.basename(os.environ.get("COLOC_HIP_LIB", "in tree")), W, H, (time.perf_counter() - t0) / 300 * 1e6))
ctx.close()
