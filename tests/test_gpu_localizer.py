"""HIPLocalizer / HIPRobustMatcher (coloc_amd/host) driven through a C++ program the way ColoC drives Localizer /
RobustMatcher (reference Localizer.hpp:59-177, RobustMatcher.hpp:153-186): radial-K3 undistortion of the query pixels,
a-contrario P3P RANSAC, refinement + covariance, the bool = failure convention; five-point AC-RANSAC + relative pose
from E with the chirality vote.  Checked against ground truth and against the Python binding of the same C ABI."""
import subprocess

import numpy as np
import pytest

import synth
from test_policy_host import build_driver
from test_gpu_epipolar import _two_view

pytestmark = pytest.mark.gpu


def _distort(x, K, k):
    """OpenMVG's radial K3 model: x_d = x_u (1 + k1 r^2 + k2 r^4 + k3 r^6) on the camera plane."""
    f, pp = K[0, 0], np.array([K[0, 2], K[1, 2]])
    c = (x - pp) / f
    r2 = (c ** 2).sum(1, keepdims=True)
    return c * (1 + r2 * (k[0] + r2 * (k[1] + r2 * k[2]))) * f + pp


def test_localizer_and_two_view_drivers(tmp_path, gpu_ctx):
    exe = build_driver(str(tmp_path / "localizer_driver"), "localizer_driver.cpp")
    # ---- localisation: 900 map points, 1500 query features, 700 tracked matches of which 30 % are wrong
    sc = synth.pnp_scene(700, seed=4321, outlier_frac=0.3)
    K, kd = sc["K"], np.array([-0.12, 0.05, -0.01])
    rng = np.random.default_rng(5)
    n_map, n_feat, n = 900, 1500, 700
    map_rows = rng.choice(n_map, n, replace=False)
    feat_rows = rng.choice(n_feat, n, replace=False)
    mapX = rng.uniform(-5, 5, (n_map, 3)) + [0, 0, 12]
    mapX[map_rows] = sc["X"]
    feats = np.stack([rng.uniform(0, 1280, n_feat), rng.uniform(0, 720, n_feat)], 1)
    feats[feat_rows] = _distort(sc["x"], K, kd)                  # the detector sees DISTORTED pixels
    feats = feats.astype(np.float32).astype(np.float64)           # features are stored as floats (SIOPointFeature)
    head = [1280, 720, K[0, 0], K[0, 2], K[1, 2], kd[0], kd[1], kd[2], n_map, n_feat, n]
    np.concatenate([head, mapX.reshape(-1), feats.reshape(-1), np.stack([map_rows, feat_rows], 1).reshape(-1)]).astype(np.float64).tofile(
        tmp_path / "loc.bin")
    # ---- two view
    x1, x2, Ftrue, out = _two_view(800, seed=31)
    K2 = synth.pnp_scene(5, seed=31)["K"]
    np.concatenate([[1280, 720, K2[0, 0], K2[0, 2], K2[1, 2], 800], x1.reshape(-1), x2.reshape(-1)]).astype(np.float64).tofile(
        tmp_path / "twoview.bin")
    res = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr

    o = np.fromfile(tmp_path / "loc_out.bin", dtype=np.float64)
    assert o[0] == 0.0                                            # EXIT_SUCCESS through bool: false = success
    R, C, cov, rmse, n_inl = o[1:10].reshape(3, 3), o[10:13], o[13:49].reshape(6, 6), o[49], int(o[50])
    inl = o[51:51 + n_inl].astype(int)
    Ctrue = -sc["R"].T @ sc["t"]
    ang = np.degrees(np.arccos(np.clip((np.trace(R @ sc["R"].T) - 1) / 2, -1, 1)))
    assert ang < 0.05 and np.linalg.norm(C - Ctrue) < 0.02        # 0.5 px noise, float32 feature storage
    good = np.zeros(n, bool); good[inl] = True
    assert (good & sc["inliers"]).sum() >= 0.9 * sc["inliers"].sum() and (good & ~sc["inliers"]).sum() <= 0.02 * n   # the a-contrario cut drops the noise tail
    assert 0.2 < rmse < 1.5 and np.allclose(cov, cov.T, rtol=1e-6, atol=1e-15) and (np.diag(cov) > 0).all()
    # the same call through the Python binding on the undistorted float pixels: identical inliers (same seed = 1)
    from solvers_np import undistort_k3
    xu = undistort_k3(feats[feat_rows], K, kd)
    ref = gpu_ctx.pnp_acransac(sc["X"], xu, K, max_iteration=256, seed=1)
    inside = (sc["x"][:, 0] > 0) & (sc["x"][:, 0] < 1280) & (sc["x"][:, 1] > 0) & (sc["x"][:, 1] < 720)
    assert np.abs(xu - sc["x"])[inside].max() < 1e-3              # the bisection inverts the distortion (float32 storage limits it)
    assert np.array_equal(np.sort(ref["inliers"]), np.sort(inl))
    # second call site (Reconstructor::resectionCamera -> SfM_Localizer::Localize directly): same solve, same seed, no refinement
    tail = o[51 + n_inl:]
    assert tail[0] == 1.0 and int(tail[1]) == n_inl and np.linalg.norm(tail[2:5] - Ctrue) < 0.05
    assert tail[5] == 0.0                                          # without intrinsics: the uncalibrated kernel is not provided
    # HIPLocalizer::localizeImages (config[2]'s batched pose): two cameras in one call == two localizeImage calls in a row, bit for bit
    assert tail[6] == 1.0 and tail[7] == 0.0

    t = np.fromfile(tmp_path / "twoview_out.bin", dtype=np.float64)
    assert t[0] == 0.0
    E, R2, C2, n2 = t[1:10].reshape(3, 3), t[10:19].reshape(3, 3), t[19:22], int(t[23])
    assert n2 >= 0.85 * (800 - len(out))
    a = synth.pnp_scene(5, seed=31, cam=0); b = synth.pnp_scene(5, seed=34, cam=3)
    Rrel = b["R"] @ a["R"].T
    trel = b["t"] - Rrel @ a["t"]
    Crel = -Rrel.T @ trel
    ang = np.degrees(np.arccos(np.clip((np.trace(R2 @ Rrel.T) - 1) / 2, -1, 1)))
    cosb = (C2 @ Crel) / (np.linalg.norm(C2) * np.linalg.norm(Crel))
    assert ang < 0.5 and cosb > 0.995 and abs(np.linalg.norm(C2) - 1) < 1e-9     # rotation, baseline DIRECTION (scale is free), unit t
    assert abs(np.linalg.det(R2) - 1) < 1e-9
    # filterMatches (what ColoC calls): same inlier count through regions + putative matches (float32 feature storage moves a few
    # borderline points), every kept match is one of the putative ones, the pose map is filled with the same baseline direction
    tail = t[24 + n2:]
    n_geo, n_pose, n_cons = int(tail[0]), int(tail[1]), int(tail[2])
    assert abs(n_geo - n2) <= 0.02 * n2 and n_pose == 1 and n_cons == n_geo
    C3 = tail[3:6]
    assert (C3 @ Crel) / (np.linalg.norm(C3) * np.linalg.norm(Crel)) > 0.995
    # an unknown model letter answers kModelNotOnGpuPath -- a status of its own, not the failure of an estimate; 'E' and 'F' leave kOk
    assert tail[6] == 1.0
