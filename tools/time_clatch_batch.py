"""bench.py's describe launch (2 cameras x 10k keypoints, 640x480) for the library COLOC_HIP_LIB points to: clatch_kernel time between
HIP events, 300 steps back to back behind 300 settling ones, + a checksum of the descriptors (bit-exact variants print the same)."""
import os, sys, time, hashlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
W, H, N = 640, 480, 10000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
ctx = Context(device=0, width=W, height=H, maxkp=N)
scene = synth.rect_image(W, H, seed=1000, noise_sigma=0.0).astype(np.float32)
base = synth.random_keypoints(N, W, H, seed=2000)
imgs = [torch.from_numpy(np.clip(scene + np.random.default_rng(1100 + c).normal(0.0, 2.0, scene.shape) + 0.5, 0, 255).astype(np.uint8)).to(dev) for c in range(2)]
kps = [torch.from_numpy(base[np.random.default_rng(2100 + c).permutation(N)].view(np.uint8).reshape(-1, 20).copy()).to(dev) for c in range(2)]
arena = torch.zeros((2, N, 64), dtype=torch.uint8, device=dev)
args = ([t.data_ptr() for t in imgs], W, H, W, [t.data_ptr() for t in kps], [N, N], [arena[0].data_ptr(), arena[1].data_ptr()], s)
for _ in range(300): ctx.describe_batch_dev(*args)
torch.cuda.synchronize()
ctx.profile_reset(); ctx.profile_enable(True, only=["clatch_kernel"])
t0 = time.perf_counter()
for _ in range(300): ctx.describe_batch_dev(*args)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 300 * 1e6
ctx.profile_enable(False)
p = ctx.profile_read()["clatch_kernel"]
print("%-24s clatch_kernel %6.2f us (events, 2 x 10k)   describe step %6.2f us   sha %s" % (os.path.basename(os.environ.get("COLOC_HIP_LIB", "in tree")),
      p[0] / p[1] * 1e3, dt, hashlib.sha256(arena.cpu().numpy().tobytes()).hexdigest()[:16]))
ctx.close()
