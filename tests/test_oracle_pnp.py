"""CPU tests of the PnP residual oracle: numpy fp64 cross-check, ground-truth pose gives pixel-noise
residuals on inliers, scoring counts."""
import numpy as np

import synth


def test_residuals_vs_numpy(oracle):
    sc = synth.pnp_scene(500, seed=4000)
    Rt = synth.random_poses(12, base_R=sc["R"], base_t=sc["t"], jitter=0.05)
    e = oracle.pnp_residuals(Rt, sc["X"], sc["x"], sc["K"])
    for h in range(12):
        P = Rt[h].reshape(3, 4)
        Xc = sc["X"] @ P[:, :3].T + P[:, 3]
        uvw = Xc @ sc["K"].T
        ref = ((sc["x"] - uvw[:, :2] / uvw[:, 2:3]) ** 2).sum(1)
        assert np.allclose(e[h], ref, rtol=1e-12, atol=1e-12)


def test_ground_truth_pose_separates_inliers(oracle):
    sc = synth.pnp_scene(1000, seed=4001)
    Rt = np.concatenate([sc["R"], sc["t"][:, None]], 1).reshape(1, 12)
    e = oracle.pnp_residuals(Rt, sc["X"], sc["x"], sc["K"])[0]
    assert np.sqrt(e[sc["inliers"]]).mean() < 1.5          # sigma = 0.5 px noise
    cnt, cost = oracle.pnp_score(e[None], 16.0)
    assert sc["inliers"].sum() <= cnt[0] <= sc["inliers"].sum() + 20
    assert cost[0] > 0
