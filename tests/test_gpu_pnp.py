"""GPU parity of the batched PnP residual / scoring kernels (fp64).  Tolerances: the residual
matrix is reproduced EXACTLY (same operation order, no FMA); the tree-reduced truncated cost is
within 1e-12 relative of the sequential sum; inlier counts are exact."""
import os

import numpy as np
import pytest

import synth
from solvers_np import numpy_p3p as _numpy_p3p

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


def test_p3p_hypotheses_vs_independent_solver(gpu_ctx):
    sc = synth.pnp_scene(400, seed=4002, outlier_frac=0.0, noise_sigma=0.0)
    rng = np.random.default_rng(3)
    samples = np.stack([rng.choice(400, 3, replace=False) for _ in range(300)]).astype(np.int32)
    hyp = gpu_ctx.pnp_p3p(sc["X"], sc["x"], sc["K"], samples)
    true = np.concatenate([sc["R"], sc["t"][:, None]], 1)
    hit = 0
    for s in range(len(samples)):
        got = [h.reshape(3, 4) for h in hyp[s] if not np.isnan(h).any()]
        want = _numpy_p3p(sc["X"][samples[s]], sc["x"][samples[s]], sc["K"])
        # every GPU pose is a pose of the independent solver.  Tolerance 1e-4 absolute on [R|t]: near
        # double roots of the quartic amplify rounding (observed worst 1.2e-5); typical agreement is 1e-10.
        for g in got:
            assert min(np.abs(g - w).max() for w in want) < 1e-4
        if got and min(np.abs(g - true).max() for g in got) < 1e-6:
            hit += 1
    assert hit >= 0.98 * len(samples)            # exact data: the true pose is among the roots


def test_ransac_recovers_pose_under_noise_and_outliers(gpu_ctx):
    for N in (200, 1000, 5000):
        sc = synth.pnp_scene(N, seed=4000 + N, outlier_frac=0.3, noise_sigma=0.5)
        Rt, mask, cost = gpu_ctx.pnp_ransac(sc["X"], sc["x"], sc["K"], n_samples=256, seed=7, thr2=16.0)
        assert Rt is not None
        R, t = Rt[:, :3], Rt[:, 3]
        ang = np.degrees(np.arccos(np.clip((np.trace(R @ sc["R"].T) - 1) / 2, -1, 1)))
        assert ang < 0.5 and np.linalg.norm(t - sc["t"]) < 0.1            # tolerance: 0.5 deg, 0.1 units at 0.5 px noise
        assert (mask & sc["inliers"]).sum() >= 0.9 * sc["inliers"].sum()
        assert (mask & ~sc["inliers"]).sum() <= 0.05 * N


def test_ransac_is_deterministic_and_matches_scored_hypotheses(gpu_ctx, oracle):
    sc = synth.pnp_scene(800, seed=4010)
    rng = np.random.default_rng(11)
    samples = np.stack([rng.choice(800, 3, replace=False) for _ in range(128)]).astype(np.int32)
    Rt1, m1, c1 = gpu_ctx.pnp_ransac(sc["X"], sc["x"], sc["K"], samples=samples, thr2=9.0)
    Rt2, m2, c2 = gpu_ctx.pnp_ransac(sc["X"], sc["x"], sc["K"], samples=samples, thr2=9.0)
    assert np.array_equal(Rt1, Rt2) and np.array_equal(m1, m2) and c1 == c2
    # the winner is the best of the GPU's own hypotheses when re-scored by the ORACLE
    hyp = gpu_ctx.pnp_p3p(sc["X"], sc["x"], sc["K"], samples).reshape(-1, 12)
    valid = ~np.isnan(hyp).any(1)
    err = oracle.pnp_residuals(hyp[valid], sc["X"], sc["x"], sc["K"])
    cnt, cost = oracle.pnp_score(err, 9.0)
    order = np.lexsort((np.arange(len(cnt)), cost, -cnt))
    assert np.array_equal(hyp[valid][order[0]].reshape(3, 4), Rt1)
    assert m1.sum() == cnt[order[0]] and np.array_equal(m1, err[order[0]] < 9.0)


def test_ransac_degenerate_inputs(gpu_ctx):
    sc = synth.pnp_scene(50, seed=1)
    Rt, mask, _ = gpu_ctx.pnp_ransac(sc["X"][:2], sc["x"][:2], sc["K"])
    assert Rt is None and not mask.any()
    X = np.zeros((10, 3)); X[:, 2] = 5.0                      # all points identical: no triad
    Rt, mask, _ = gpu_ctx.pnp_ransac(X, np.full((10, 2), 100.0), sc["K"])
    assert Rt is None or np.isfinite(Rt).all()


# ---- refinement: numpy restatement of the same cost as the test-side checker ----------

def _rodrigues(w):
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th ** 2 * Kx @ Kx


def _log_so3(R):
    th = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return v / 2 if th < 1e-9 else th / (2 * np.sin(th)) * v


def _residuals(p, X, x, K):
    Xc = X @ _rodrigues(p[:3]).T + p[3:]
    uvw = Xc @ K.T
    return x - uvw[:, :2] / uvw[:, 2:3]


def _huber_cost(p, X, x, K, a=16.0):
    s = (_residuals(p, X, x, K) ** 2).sum(1)
    return 0.5 * np.where(s <= a * a, s, 2 * a * np.sqrt(s) - a * a).sum()


def _numeric_cov(p, X, x, K, a=16.0):
    eps = 1e-6
    J = np.zeros((len(X), 2, 6))
    for k in range(6):
        d = np.zeros(6); d[k] = eps
        J[:, :, k] = (_residuals(p + d, X, x, K) - _residuals(p - d, X, x, K)) / (2 * eps)
    s = (_residuals(p, X, x, K) ** 2).sum(1)
    w = np.where(s <= a * a, 1.0, a / np.sqrt(np.maximum(s, 1e-300)))
    A = np.einsum("n,nij,nik->jk", w, J, J)
    return np.linalg.inv(A)


def test_refine_reaches_the_minimum_and_covariance_matches(gpu_ctx):
    from scipy.optimize import minimize
    for N, seed in ((300, 1), (2000, 2)):
        sc = synth.pnp_scene(N, seed=4200 + seed, outlier_frac=0.2, noise_sigma=0.5)
        Rt0, mask, _ = gpu_ctx.pnp_ransac(sc["X"], sc["x"], sc["K"], n_samples=256, seed=3, thr2=16.0)
        Xi, xi = sc["X"][mask], sc["x"][mask]
        Rt, cov, rmse, it = gpu_ctx.pnp_refine(sc["X"], sc["x"], sc["K"], Rt0, mask=mask)
        p = np.concatenate([_log_so3(Rt[:, :3]), Rt[:, 3]])
        p0 = np.concatenate([_log_so3(Rt0[:, :3]), Rt0[:, 3]])
        c_gpu, c_0 = _huber_cost(p, Xi, xi, sc["K"]), _huber_cost(p0, Xi, xi, sc["K"])
        ref = minimize(_huber_cost, p0, args=(Xi, xi, sc["K"]), method="BFGS", options={"gtol": 1e-10})
        assert c_gpu <= c_0 and c_gpu <= ref.fun * (1 + 1e-9) + 1e-9          # at least as good as scipy's minimum
        assert np.abs(p - ref.x).max() < 1e-5                                  # same minimiser (tolerance 1e-5 in [w|t])
        assert np.isclose(rmse, np.sqrt(c_gpu / (2 * mask.sum())), rtol=1e-9)
        cov_ref = _numeric_cov(p, Xi, xi, sc["K"])
        assert np.allclose(cov, cov_ref, rtol=1e-4, atol=1e-12)               # analytic vs central-difference Jacobian
        err = lambda M: np.linalg.norm(M[:, 3] - sc["t"])
        assert err(Rt) <= err(Rt0) + 1e-9                                      # closer to the truth than the minimal-sample pose
        assert 1 <= it <= 50


def test_refine_with_huber_tail_and_without_mask(gpu_ctx):
    """No inlier mask and 20 % gross outliers: the Huber tail (16 px) keeps the minimum near the truth."""
    sc = synth.pnp_scene(1500, seed=4300, outlier_frac=0.2, noise_sigma=0.5)
    Rt0 = np.concatenate([sc["R"], sc["t"][:, None]], 1)
    Rt0[:, 3] += 0.05
    Rt, cov, rmse, it = gpu_ctx.pnp_refine(sc["X"], sc["x"], sc["K"], Rt0, mask=None)
    p = np.concatenate([_log_so3(Rt[:, :3]), Rt[:, 3]])
    p0 = np.concatenate([_log_so3(Rt0[:, :3]), Rt0[:, 3]])
    c = _huber_cost(p, sc["X"], sc["x"], sc["K"])
    assert c < _huber_cost(p0, sc["X"], sc["x"], sc["K"])
    g = np.array([(_huber_cost(p + e, sc["X"], sc["x"], sc["K"]) - _huber_cost(p - e, sc["X"], sc["x"], sc["K"])) / 2e-6
                  for e in np.eye(6) * 1e-6])
    assert np.abs(g).max() < 1e-5 * c                                          # stationary point of the robust cost
    assert np.all(np.linalg.eigvalsh(cov) > 0)


def test_localize_equals_ransac_then_refine(gpu_ctx):
    sc = synth.pnp_scene(1200, seed=4400)
    rng = np.random.default_rng(2)
    samples = np.stack([rng.choice(1200, 3, replace=False) for _ in range(256)]).astype(np.int32)
    Rt0, mask0, _ = gpu_ctx.pnp_ransac(sc["X"], sc["x"], sc["K"], samples=samples, thr2=16.0)
    Rt1, cov1, rmse1, _ = gpu_ctx.pnp_refine(sc["X"], sc["x"], sc["K"], Rt0, mask=mask0)
    Rt2, cov2, mask2, rmse2 = gpu_ctx.pnp_localize(sc["X"], sc["x"], sc["K"], samples=samples, thr2=16.0)
    assert np.array_equal(mask0, mask2) and np.array_equal(Rt1, Rt2) and np.array_equal(cov1, cov2) and rmse1 == rmse2
    # no pose -> zeros, no crash
    Rt, cov, mask, rmse = gpu_ctx.pnp_localize(sc["X"][:2], sc["x"][:2], sc["K"])
    assert Rt is None and not mask.any() and not cov.any()
