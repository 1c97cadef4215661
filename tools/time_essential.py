import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from coloc_amd import Context
ctx = Context(device=0, width=640, height=480, maxkp=1024)
rng = np.random.default_rng(1)
N = 1000
X = np.stack([rng.uniform(-5, 5, N), rng.uniform(-5, 5, N), rng.uniform(4, 20, N)], 1)
K = np.array([[1000.0, 0, 640], [0, 1000.0, 360], [0, 0, 1]])
a = 0.2; R = np.array([[np.cos(a), 0, -np.sin(a)], [0, 1, 0], [np.sin(a), 0, np.cos(a)]]); t = np.array([0.5, 0.1, 0.2])
x1 = X @ K.T; x1 = x1[:, :2] / x1[:, 2:3]
Xc = X @ R.T + t; x2 = Xc @ K.T; x2 = x2[:, :2] / x2[:, 2:3]
x2 += rng.normal(0, 0.5, x2.shape)
out_idx = rng.choice(N, 300, replace=False); x2[out_idx] = np.stack([rng.uniform(0, 1280, 300), rng.uniform(0, 720, 300)], 1)
ts = []
for it in range(40):
    t0 = time.perf_counter(); E, F, mask = ctx.essential_ransac(x1, x2, K, K, n_samples=256, seed=it + 1, thr2=4.0); ts.append(time.perf_counter() - t0)
print("essential_ransac N=1000 S=256: p50 %.1f us, inliers %d" % (np.sort(ts[5:])[17] * 1e6, mask.sum()))
ctx.close()
