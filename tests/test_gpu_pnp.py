"""GPU parity of the batched PnP residual / scoring kernels (fp64).  Tolerances: the residual
matrix is reproduced EXACTLY (same operation order, no FMA); the tree-reduced truncated cost is
within 1e-12 relative of the sequential sum; inlier counts are exact."""
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_golden(gpu_ctx):
    g = np.load(os.path.join(G, "pnp_8x64.npz"))
    assert np.array_equal(gpu_ctx.pnp_residuals(g["Rt"], g["X"], g["x"], g["K"]), g["err"])
    cnt, cost = gpu_ctx.pnp_score(g["Rt"], g["X"], g["x"], g["K"], 16.0)
    assert np.array_equal(cnt, g["count"]) and np.allclose(cost, g["cost"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("H,N", [(1, 1), (3, 5), (64, 200), (256, 1000), (1024, 5000), (1024, 257)])
def test_residuals_exact(gpu_ctx, oracle, H, N):
    sc = synth.pnp_scene(N, seed=4000 + N)
    Rt = synth.random_poses(H, seed=4100 + H, base_R=sc["R"], base_t=sc["t"], jitter=0.05)
    e = gpu_ctx.pnp_residuals(Rt, sc["X"], sc["x"], sc["K"])
    eo = oracle.pnp_residuals(Rt, sc["X"], sc["x"], sc["K"])
    assert np.array_equal(e, eo)
    cnt, cost = gpu_ctx.pnp_score(Rt, sc["X"], sc["x"], sc["K"], 16.0)
    cnt_o, cost_o = oracle.pnp_score(eo, 16.0)
    assert np.array_equal(cnt, cnt_o)
    assert np.allclose(cost, cost_o, rtol=1e-12, atol=0)


def test_ground_truth_pose_wins(gpu_ctx):
    sc = synth.pnp_scene(2000, seed=4001)
    Rt = synth.random_poses(512, seed=1, base_R=sc["R"], base_t=sc["t"], jitter=0.05)
    Rt[137] = np.concatenate([sc["R"], sc["t"][:, None]], 1).reshape(-1)
    cnt, cost = gpu_ctx.pnp_score(Rt, sc["X"], sc["x"], sc["K"], 16.0)
    assert cnt.argmax() == 137 and cost.argmin() == 137
    assert cnt[137] >= sc["inliers"].sum()
