"""clatch_kernel time for one camera x N keypoints (N = 2500 / 5000 / 10000 / 20000; 640x480), events around the launch, 300 back to back."""
import os, sys, time, hashlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
W, H = 640, 480
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
ctx = Context(device=0, width=W, height=H, maxkp=20000)
img = synth.rect_image(W, H, seed=1000, noise_sigma=2.0)
ctx.pyramid_build(img)
out = []
for N in (2500, 5000, 10000, 20000):
    kps = synth.random_keypoints(N, W, H, seed=2000)
    dk = torch.from_numpy(kps.view(np.uint8).reshape(-1, 20).copy()).to(dev)
    dd = torch.empty((N, 64), dtype=torch.uint8, device=dev)
    for _ in range(300): ctx.describe_dev(dk.data_ptr(), N, dd.data_ptr(), s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): ctx.describe_dev(dk.data_ptr(), N, dd.data_ptr(), s)
    torch.cuda.synchronize()
    out.append("%d: %.2f us (%s)" % (N, (time.perf_counter() - t0) / 300 * 1e6, hashlib.sha256(dd.cpu().numpy().tobytes()).hexdigest()[:8]))
print("%-14s %s" % (os.path.basename(os.environ.get("COLOC_HIP_LIB", "in tree")), "   ".join(out)))
ctx.close()
