#!/usr/bin/env python3
"""Soak of clc_describe_match_pair_dev: random keypoint counts of both cameras (0 .. 10 000, ragged against the 256-row query blocks and
the 2 048-row progress groups), random chunkings (0, 1, 2 .. 16), both sweep formulations, several steps back to back on ONE context;
descriptors == oracle CLATCH, matches == oracle K2NN, every time.  usage: soak_pair_step.py [cases]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth, oracle_lib
from coloc_amd import Context
W, H = 640, 480
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
orc = oracle_lib.Oracle()
rng = np.random.default_rng(7)
ctx = Context(device=0, width=W, height=H, maxkp=10000)
scene = synth.rect_image(W, H, seed=1000, noise_sigma=0.0).astype(np.float32)
imgs = [np.clip(scene + np.random.default_rng(5 + c).normal(0.0, 2.0, scene.shape) + 0.5, 0, 255).astype(np.uint8) for c in range(2)]
pyr = [orc.pyramid(i) for i in imgs]
base = synth.random_keypoints(10000, W, H, seed=2000)
kps = [base[np.random.default_rng(11 + c).permutation(10000)] for c in range(2)]
full = [orc.clatch(pyr[c], kps[c]) for c in range(2)]
d_imgs = [torch.from_numpy(i).cuda() for i in imgs]
d_kps = [torch.from_numpy(k.view(np.uint8).reshape(-1, 20).copy()).cuda() for k in kps]
t0 = time.time()
edges = [0, 1, 255, 256, 257, 2047, 2048, 2049, 4096, 8191, 8192, 8193, 9999, 10000]
for it in range(cases):
    nq = int(rng.choice(edges)) if rng.random() < 0.4 else int(rng.integers(0, 10001))
    nt = int(rng.choice(edges)) if rng.random() < 0.3 else int(rng.integers(0, 10001))
    chunks = int(rng.choice([0, 1, 2, 3, 4, 5, 7, 16]))
    form = ("matrix", "popcount")[it % 5 == 4]
    thr = int(rng.integers(0, 100))
    ctx.set_k2nn_formulation(form)
    want = orc.k2nn(full[0][:nq], full[1][:nt], thr) if nq and nt else np.full(nq, -1, np.int32)
    desc = [torch.full((10000, 64), 0x5A, dtype=torch.uint8, device="cuda") for _ in range(2)]
    match = torch.full((10000,), -7, dtype=torch.int32, device="cuda")
    for rep in range(int(rng.integers(1, 4))):
        ctx.describe_match_pair_dev([t.data_ptr() for t in d_imgs], W, H, W, [t.data_ptr() for t in d_kps], [nq, nt],
                                    [t.data_ptr() for t in desc], thr, match.data_ptr(), chunks=chunks)
    ctx.sync()
    assert np.array_equal(desc[0].cpu().numpy()[:nq], full[0][:nq]) and np.array_equal(desc[1].cpu().numpy()[:nt], full[1][:nt]), (it, nq, nt, chunks, form)
    assert (desc[0].cpu().numpy()[nq:] == 0x5A).all() and (desc[1].cpu().numpy()[nt:] == 0x5A).all(), (it, "rows past the count were written")
    assert np.array_equal(match.cpu().numpy()[:nq], want), (it, nq, nt, chunks, form, thr)
    if it % 10 == 0:
        print("%d cases ok (%.0f s)" % (it + 1, time.time() - t0), flush=True)
print("soak ok: %d pair steps" % cases)
ctx.close()
