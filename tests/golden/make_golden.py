#!/usr/bin/env python3
"""Generate the committed golden fixtures (small .npz files) from the CPU oracle and, for the
feeder, from the REFERENCE's own KFAST.h / FeatureAngle.h compiled into oracle/_ref (this only
works in the build container, where /root/reference exists).

The reference ships no golden vectors for this path, so for the three kernels
these fixtures freeze the restatement's outputs (regression pins); `feeder_*` arrays are genuine
reference outputs (keypoints + angles produced by the reference's code run here).

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import synth  # noqa: E402


def main():
    orc = oracle_lib.Oracle()
    # --- K2NN: 2 x 512 descriptors, thresholds 40 / 60, with planted duplicates
    Q, T = synth.planted_descriptors(512, 512, seed=3000)
    T[400] = T[7]
    Q[5] = T[7]
    m40, b, s = orc.k2nn(Q, T, 40, want_dist=True)
    m60 = orc.k2nn(Q, T, 60)
    np.savez_compressed(os.path.join(HERE, "k2nn_512.npz"), Q=Q, T=T, match40=m40, match60=m60, best=b, second=s)
    # --- pyramid + CLATCH: 160x120 image, 64 keypoints over all 8 levels incl. borders and special angles
    W, H = 160, 120
    img = synth.rect_image(W, H, n_rect=60, seed=1000, noise_sigma=2.0)
    pyr = orc.pyramid(img)
    kps = synth.random_keypoints(64, W, H, seed=2000)
    ws, hs, _ = synth.pyramid_dims(W, H)
    special = [0.0, np.pi / 2, -np.pi / 2, np.pi, -np.pi, np.pi / 4, 1e-3, -3.0]
    for i, a in enumerate(special):
        kps["angle"][i] = np.float32(a)
    for i in range(8):            # one keypoint per level pinned to a corner region (clamp path)
        kps["scale"][8 + i] = i
        kps["x"][8 + i] = 3 if i % 2 == 0 else ws[i] - 4
        kps["y"][8 + i] = 3 if i % 3 == 0 else hs[i] - 4
    desc = orc.clatch(pyr, kps)
    np.savez_compressed(os.path.join(HERE, "clatch_160x120.npz"), img=img, kps=kps.view(np.uint8).reshape(-1, 20),
                        desc=desc, **{"level%d" % i: pyr[i] for i in range(8)})
    # --- PnP residuals: 8 hypotheses x 64 points
    sc = synth.pnp_scene(64, seed=4000)
    Rt = synth.random_poses(8, base_R=sc["R"], base_t=sc["t"], jitter=0.02)
    Rt[0] = np.concatenate([sc["R"], sc["t"][:, None]], 1).reshape(-1)
    err = orc.pnp_residuals(Rt, sc["X"], sc["x"], sc["K"])
    cnt, cost = orc.pnp_score(err, 16.0)
    np.savez_compressed(os.path.join(HERE, "pnp_8x64.npz"), Rt=Rt, X=sc["X"], x=sc["x"], K=sc["K"], err=err, count=cnt, cost=cost)
    # --- feeder: reference KFAST + featureAngle outputs on a 214x161 image (width with the shift quirk) and a 320x240 one
    try:
        ref = oracle_lib.RefFeeder()
    except (FileNotFoundError, OSError):
        print("oracle/_ref missing: feeder fixture not regenerated")
        return
    out = {}
    for name, (w, h, seed) in {"a": (214, 161, 5), "b": (320, 240, 6)}.items():
        im = synth.rect_image(w, h, n_rect=100, seed=seed, noise_sigma=2.0)
        k = ref.kfast(im, 40, True)
        ang = np.array([ref.feature_angle(im, int(p["x"]), int(p["y"])) for p in k], dtype=np.float32)
        out["img_" + name] = im
        out["xys_" + name] = np.stack([k["x"], k["y"], k["score"].astype(np.int32)], 1)
        out["angle_" + name] = ang
    np.savez_compressed(os.path.join(HERE, "feeder_ref.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
