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


def _true_E(seed):
    a = synth.pnp_scene(5, seed=seed, cam=0)
    b = synth.pnp_scene(5, seed=seed + 3, cam=3)
    R = b["R"] @ a["R"].T
    t = b["t"] - R @ a["t"]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    return tx @ R


def _numpy_fivepoint_check(E, q1, q2):
    """Necessary and sufficient conditions of a five-point solution: the 5 epipolar constraints, det E = 0 and
    2 E E^T E - tr(E E^T) E = 0 (two equal singular values, one zero)."""
    En = E / np.linalg.norm(E)
    r = np.abs(np.einsum("ni,ij,nj->n", np.c_[q2, np.ones(5)], En, np.c_[q1, np.ones(5)])).max()
    s = np.linalg.svd(En, compute_uv=False)
    return r, abs(s[0] - s[1]) / s[0], s[2] / s[0]


def test_fivepoint_hypotheses_are_valid_and_contain_the_truth(gpu_ctx):
    N = 400
    x1, x2, F, _ = _two_view(N, seed=91, noise=0.0, outliers=0.0)
    K = synth.pnp_scene(5, seed=1)["K"]
    Kinv = np.linalg.inv(K)
    rng = np.random.default_rng(4)
    samples = np.stack([rng.choice(N, 5, replace=False) for _ in range(200)]).astype(np.int32)
    Es = gpu_ctx.essential_fivepoint(x1, x2, K, K, samples)
    Et = _true_E(91)
    Et /= np.linalg.norm(Et)
    hits, nsol = 0, 0
    for s in range(len(samples)):
        q1 = (np.c_[x1[samples[s]], np.ones(5)] @ Kinv.T)[:, :2]
        q2 = (np.c_[x2[samples[s]], np.ones(5)] @ Kinv.T)[:, :2]
        best = 1.0
        for E in Es[s]:
            if np.isnan(E).any():
                continue
            nsol += 1
            E = E.reshape(3, 3)
            r, ds, s3 = _numpy_fivepoint_check(E, q1, q2)
            # a genuine essential matrix through the 5 points; 1e-4 on the singular values: roots of the cubic
            # constraints that are close together are only resolved to ~1e-5 (observed worst 4.4e-6)
            assert r < 1e-9 and ds < 1e-4 and s3 < 1e-4
            En = E / np.linalg.norm(E)
            best = min(best, np.abs(En - Et).max(), np.abs(En + Et).max())
        hits += best < 1e-5
    assert hits >= 0.95 * len(samples) and 2.0 < nsol / len(samples) <= 10.0


def test_essential_ransac_finds_the_model(gpu_ctx):
    for N in (300, 3000):
        x1, x2, F, out = _two_view(N, seed=60 + N, noise=0.5, outliers=0.3)
        K = synth.pnp_scene(5, seed=1)["K"]
        E, Fh, mask = gpu_ctx.essential_ransac(x1, x2, K, K, n_samples=256, seed=5, thr2=4.0)
        assert E is not None
        inl = np.ones(N, bool); inl[out] = False
        assert (mask & inl).sum() >= 0.85 * inl.sum() and (mask & ~inl).sum() <= 0.06 * N
        s = np.linalg.svd(E, compute_uv=False)
        assert abs(s[0] - s[1]) / s[0] < 1e-4 and s[2] / s[0] < 1e-4
        # same epipolar geometry as the ground truth: the true F's inliers are (almost all) ours
        Ft = F / np.linalg.norm(F)
        e_true = gpu_ctx.epipolar_residuals(Ft.reshape(1, 9), x1, x2)[0]
        assert ((e_true < 4.0) & mask).sum() >= 0.9 * (e_true < 4.0).sum()
    E, Fh, mask = gpu_ctx.essential_ransac(x1[:4], x2[:4], K, K)
    assert E is None and not mask.any()


def test_wave_solver_agrees_with_the_sequential_statement(gpu_ctx):
    """csrc/fivept_wave.h (one problem per wave, lanes sharing the work) against csrc/fivept.h run on the host: the same
    algorithm step for step, so the solution SETS agree -- not bit for bit (the wave form sums the Gauss-Newton rows in pairs
    and fuses multiply-adds), but to 1e-7 on unit-norm matrices for all but a few ill-conditioned samples."""
    import fivept_host
    N = 600
    x1, x2, _, _ = _two_view(N, seed=33, noise=0.3, outliers=0.2)
    K = synth.pnp_scene(5, seed=1)["K"]
    Kinv = np.linalg.inv(K)
    rng = np.random.default_rng(8)
    samples = np.stack([rng.choice(N, 5, replace=False) for _ in range(300)]).astype(np.int32)
    Es = gpu_ctx.essential_fivepoint(x1, x2, K, K, samples)
    same = 0
    for s in range(len(samples)):
        q1 = (np.c_[x1[samples[s]], np.ones(5)] @ Kinv.T)[:, :2]
        q2 = (np.c_[x2[samples[s]], np.ones(5)] @ Kinv.T)[:, :2]
        host = [E / np.linalg.norm(E) for E in fivept_host.solve(q1, q2)]
        dev = [E.reshape(3, 3) / np.linalg.norm(E) for E in Es[s] if not np.isnan(E).any()]
        ok = len(host) == len(dev)
        if ok:
            for E in dev:
                ok = ok and any(min(np.abs(E - Hh).max(), np.abs(E + Hh).max()) < 1e-7 for Hh in host)
        same += ok
    assert same >= 0.97 * len(samples), same / len(samples)
