"""Soak of the GPU detector against the oracle (which is pinned to the compiled reference): random sizes around the 64 x 16 tile seams and
the 16-column walk groups (every width class mod 16, widths 6 mod 16 over-represented), random level counts, thresholds and image kinds.
usage: soak_detect.py [cases] [seed]   -- prints the first mismatch and exits 1."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import synth, oracle_lib
from coloc_amd import Context
from test_gpu_detect import oracle_detect, same_kps
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = oracle_lib.Oracle()
done = 0
for case in range(n_cases):
    W = int(rng.integers(8, 900)); H = int(rng.integers(8, 260))
    if rng.random() < 0.4:
        W = 38 + 16 * int(rng.integers(0, 50))                     # widths 6 (mod 16): the walk replay
    levels = int(rng.integers(1, 9)); thr = int(rng.choice([5, 20, 40, 80, 200]))
    if W / 1.2 ** (levels - 1) < 8 or H / 1.2 ** (levels - 1) < 8:
        levels = 1
    kind = int(rng.integers(0, 4))
    if kind == 0 and (W < 24 or H < 24):
        kind = 1                                                     # (synth.rect_image needs room for its rectangles)
    if kind == 0:
        img = synth.rect_image(W, H, n_rect=max(4, W * H // 2500), seed=int(rng.integers(1 << 30)), noise_sigma=float(rng.choice([0.0, 2.0, 8.0])))
    elif kind == 1:
        img = rng.integers(0, 256, size=(H, W)).astype(np.uint8)
    elif kind == 2:
        yy, xx = np.mgrid[0:H, 0:W]
        img = (((yy // int(rng.integers(3, 9)) + xx // int(rng.integers(3, 9))) % 2) * int(rng.integers(60, 200)) + rng.integers(0, 30, size=(H, W))).astype(np.uint8)
    else:
        img = rng.integers(100, 120, size=(H, W)).astype(np.uint8)
        m = rng.random((H, W)); img[m < 0.03] = 255; img[m > 0.97] = 0
    ctx = Context(device=0, width=W, height=H, maxkp=400000, scale_levels=levels, fast_thresh=thr, matcher=False)
    ctx.pyramid_build(img)
    kps, found = ctx.detect()
    _, want = oracle_detect(orc, img, thresh=thr, levels=levels)
    ok = found == len(want) and same_kps(kps, want)
    ctx.close()
    if not ok:
        print("MISMATCH case %d: W %d H %d levels %d thr %d kind %d: gpu %d / oracle %d keypoints" % (case, W, H, levels, thr, kind, found, len(want)))
        np.save(os.path.join(R, "gpurun_out", "soak_detect_fail.npy"), img)
        sys.exit(1)
    done += 1
    if case % 25 == 24:
        print("%d cases ok" % done, flush=True)
print("soak ok: %d cases" % done)
