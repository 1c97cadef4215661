#!/usr/bin/env python3
"""Sustained time of the pyramid launch (one and two cameras, 640x480 and 1280x720): back-to-back launches, host clock / reps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
for W, H in [(640, 480), (1280, 720)]:
    ctx = Context(device=0, width=W, height=H, maxkp=1000)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
    imgs = [torch.from_numpy(synth.rect_image(W, H, seed=1000 + c, noise_sigma=2.0)).to(dev) for c in range(2)]
    kps = [torch.zeros((1, 20), dtype=torch.uint8, device=dev) for _ in range(2)]
    desc = [torch.zeros((1, 64), dtype=torch.uint8, device=dev) for _ in range(2)]
    def one(): ctx.pyramid_build_dev(imgs[0].data_ptr(), W, H, W, s)
    def two(): ctx.describe_batch_dev([t.data_ptr() for t in imgs], W, H, W, [t.data_ptr() for t in kps], [0, 0], [t.data_ptr() for t in desc], s)
    for name, fn in [("1 camera", one), ("2 cameras (batched, no keypoints)", two)]:
        for _ in range(20): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300): fn()
        torch.cuda.synchronize()
        print("%dx%d %-36s %.2f us per launch" % (W, H, name, (time.perf_counter() - t0) / 300 * 1e6))
    ctx.close()
