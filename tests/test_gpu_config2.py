"""BASELINE config[2] as a parity case: 4 cameras, 1280x720 synthetic scene, GPU-resident
detect + describe per camera, all 6 pairs matched in one launch group, batched PnP scoring + robust
pose on a synthetic 2D-3D problem.  Everything is checked against the oracle bit for bit (integer /
byte work) or exactly / within the stated tolerance (fp64 residuals, pose)."""
import numpy as np
import pytest

import synth
from coloc_amd import multicam
from test_gpu_detect import oracle_detect, same_kps

pytestmark = pytest.mark.gpu


def test_config2_four_cameras_all_pairs_and_pose(oracle):
    from coloc_amd import Context
    W, H, ncam = 1280, 720, 4
    ctx = Context(device=0, width=W, height=H, maxkp=30000)
    descs = []
    for c in range(ncam):
        img = synth.rect_image(W, H, seed=1000 + c // 2, noise_sigma=2.0 + c)     # cameras 0/1 and 2/3 see similar content
        kps, d, found = ctx.detect_and_describe(img)
        pyr, want = oracle_detect(oracle, img)
        assert same_kps(kps, want) and found == len(want) and len(want) > 3000
        assert np.array_equal(d, oracle.clatch(pyr, want))
        descs.append(d)
    pairs = multicam.exhaustive_pairs(ncam)
    assert len(pairs) == 6
    got = ctx.match_pairs(descs, pairs, 40)
    n_acc = 0
    for p, m in zip(pairs, got):
        want = oracle.k2nn(descs[p[0]], descs[p[1]], 40)
        assert np.array_equal(m, want)
        n_acc += int((want >= 0).sum())
    assert n_acc > 500
    # pose: N = 5000 matches, 256 samples -> <= 1024 hypotheses scored at once
    sc = synth.pnp_scene(5000, seed=4000, cam=2)
    rng = np.random.default_rng(5)
    samples = np.stack([rng.choice(5000, 3, replace=False) for _ in range(256)]).astype(np.int32)
    hyp = ctx.pnp_p3p(sc["X"], sc["x"], sc["K"], samples).reshape(-1, 12)
    valid = ~np.isnan(hyp).any(1)
    err = ctx.pnp_residuals(hyp[valid], sc["X"], sc["x"], sc["K"])
    assert np.array_equal(err, oracle.pnp_residuals(hyp[valid], sc["X"], sc["x"], sc["K"]))      # exact
    Rt, mask, _ = ctx.pnp_ransac(sc["X"], sc["x"], sc["K"], samples=samples, thr2=16.0)
    ang = np.degrees(np.arccos(np.clip((np.trace(Rt[:, :3] @ sc["R"].T) - 1) / 2, -1, 1)))
    assert ang < 0.2 and np.linalg.norm(Rt[:, 3] - sc["t"]) < 0.05          # tolerance: 0.2 deg / 0.05 units at 0.5 px noise
    assert (mask & sc["inliers"]).sum() >= 0.95 * sc["inliers"].sum()
    ctx.close()
