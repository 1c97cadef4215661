import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, synth
from coloc_amd import Context
ctx = Context(device=0, width=640, height=480, maxkp=10000)
Q, T = synth.planted_descriptors(10000, 10000, seed=1)
for _ in range(5): ctx.match_2nn(Q, T, 40)
ts = []
for _ in range(50):
    t0 = time.perf_counter(); ctx.match_2nn(Q, T, 40); ts.append(time.perf_counter() - t0)
ts = np.sort(ts)
print("clc_match_2nn host buffers 10k x 10k: p50 %.1f us -> %.0f Mmatches/s" % (ts[25] * 1e6, 1e8 / ts[25] / 1e6))
img = synth.rect_image(640, 480, seed=1000, noise_sigma=2.0); kps = synth.random_keypoints(10000, 640, 480, seed=2000)
ctx.pyramid_build(img)
for _ in range(3): ctx.describe(kps)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); ctx.pyramid_build(img); ctx.describe(kps); ts.append(time.perf_counter() - t0)
ts = np.sort(ts)
print("pyramid_build + describe host buffers 10k kp: p50 %.1f us -> %.1f Mdesc/s" % (ts[15] * 1e6, 1e4 / ts[15] / 1e6))
ts = []
for _ in range(30):
    t0 = time.perf_counter(); ctx.detect_and_describe(img); ts.append(time.perf_counter() - t0)
print("detect_and_describe host in/out 640x480: p50 %.1f us" % (np.sort(ts)[15] * 1e6))
