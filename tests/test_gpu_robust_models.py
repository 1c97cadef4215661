"""HIPRobustMatcher under the models 'F' and 'H' (coloc_amd/host/HIPRobustMatcher.hpp; reference include/coloc/RobustMatcher.hpp:
128-151 filterFundamental, :188-239 filterHomography, :106-126 decomposeHomography, :39-104 performChiralityTest, dispatch :399-405),
driven through a C++ program the way ColoC drives RobustMatcher, checked against the Python binding of the same C ABI (same seed:
same inliers, same matrix) and against the scene: the homography's motions contain the scene's (R, t / |t|), the chirality vote
picks it."""
import subprocess

import numpy as np
import pytest

import twoview_host as tvh
from test_policy_host import build_driver

pytestmark = pytest.mark.gpu


def _read(o, pos):
    status, M, R, C, prec, n = o[pos], o[pos + 1:pos + 10].reshape(3, 3), o[pos + 10:pos + 19].reshape(3, 3), o[pos + 19:pos + 22], o[pos + 22], int(o[pos + 23])
    inl = o[pos + 24:pos + 24 + n].astype(np.int32)
    return dict(status=status, M=M, R=R, C=C, precision=prec, inliers=inl), pos + 24 + n


def test_fundamental_and_homography_members(tmp_path, gpu_ctx):
    exe = build_driver(str(tmp_path / "robust_models_driver"), "robust_models_driver.cpp")
    K = tvh.K_DEFAULT
    gen = tvh.scene(800, 301, planar=False)
    pla = tvh.scene(800, 302, planar=True)
    head = [1280, 720, K[0, 0], K[0, 2], K[1, 2], 800]
    # a pure rotation as a homography: K R K^-1
    th = 0.1
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    Hrot = K @ Rz @ np.linalg.inv(K)
    np.concatenate([head, gen["x1"].reshape(-1), gen["x2"].reshape(-1)]).astype(np.float64).tofile(tmp_path / "general.bin")
    np.concatenate([head, pla["x1"].reshape(-1), pla["x2"].reshape(-1), (2.5 * pla["H"]).reshape(-1), Hrot.reshape(-1)]).astype(np.float64).tofile(tmp_path / "planar.bin")
    res = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    o = np.fromfile(tmp_path / "models_out.bin", dtype=np.float64)

    # ---- 'F'
    f, pos = _read(o, 0)
    ref = gpu_ctx.two_view_acransac("F", gen["x1"], gen["x2"], gen["wh"], max_iteration=256, seed=1)
    assert f["status"] == 0.0                                                    # EXIT_SUCCESS through bool
    assert np.array_equal(f["inliers"], ref["inliers"]) and np.array_equal(f["M"], ref["M"])
    assert f["precision"] == 5.0                                                 # RobustMatcher.hpp:145
    assert np.array_equal(f["R"], np.eye(3)) and not f["C"].any()                # no pose comes out of this model
    assert np.abs(tvh.unit(f["M"]) - tvh.unit(gen["F"])).max() < 0.05
    true_in = np.ones(800, bool); true_in[gen["outliers"]] = False
    got = np.zeros(800, bool); got[f["inliers"]] = True
    assert (got & true_in).sum() >= 0.9 * true_in.sum()
    n_geo, n_cons, n_pose, ok = o[pos:pos + 4]; pos += 4
    assert abs(n_geo - len(f["inliers"])) <= 0.03 * len(f["inliers"]) and n_cons == n_geo and n_pose == 1 and ok == 1.0   # (float32 feature storage)
    assert o[pos] == 1.0 and o[pos + 1] < 2.5 * 7; pos += 2                      # 12 matches cannot carry 17.5 inliers: EXIT_FAILURE

    # ---- 'H'
    h, pos = _read(o, pos)
    ref = gpu_ctx.two_view_acransac("H", pla["x1"], pla["x2"], pla["wh"], max_iteration=256, seed=1)
    assert h["status"] == 0.0
    assert np.array_equal(h["inliers"], ref["inliers"]) and np.array_equal(h["M"], ref["M"]) and h["precision"] == ref["error_max"]
    true_in = np.ones(800, bool); true_in[pla["outliers"]] = False
    got = np.zeros(800, bool); got[h["inliers"]] = True
    assert (got & true_in).sum() >= 0.9 * true_in.sum() and (got & ~true_in).sum() <= 3
    # The pose: the reference hands every motion's normalised translation to Pose3 in the place of the CENTRE (RobustMatcher.hpp:123) and
    # lets performChiralityTest vote on those poses (:39-104) -- restated here in numpy on the candidates the driver dumps: triangulate
    # every inlier's bearing vectors under each candidate (DLT), count the points in front of both cameras, the first maximum wins.
    assert o[pos] == 4.0; pos += 1
    cand = o[pos:pos + 48].reshape(4, 12); pos += 48
    Ki = np.linalg.inv(K)
    f1 = np.c_[pla["x1"], np.ones(800)] @ Ki.T
    f2 = np.c_[pla["x2"], np.ones(800)] @ Ki.T
    votes = []
    for m in cand:
        R, C = m[:9].reshape(3, 3), m[9:]
        t = -R @ C
        P2 = np.c_[R, t]
        cnt = 0
        for i in h["inliers"]:
            D = np.array([[-f1[i, 2], 0, f1[i, 0], 0], [0, -f1[i, 2], f1[i, 1], 0],
                          f2[i, 0] * P2[2] - f2[i, 2] * P2[0], f2[i, 1] * P2[2] - f2[i, 2] * P2[1]])
            X = np.linalg.svd(D)[2][-1]
            X = X[:3] / X[3]
            cnt += (f1[i] @ X > 0) and (f2[i] @ (R @ X + t) > 0)
        votes.append(cnt)
    pick = int(np.argmax(votes))
    assert max(votes) > 0 and np.array_equal(h["R"], cand[pick][:9].reshape(3, 3)) and np.array_equal(h["C"], cand[pick][9:]), votes
    # and the candidates hold the scene's motion: its rotation with +-(t / |t|)
    tdir = pla["t"] / np.linalg.norm(pla["t"])
    assert min(np.degrees(np.arccos(np.clip((np.trace(m[:9].reshape(3, 3) @ pla["R"].T) - 1) / 2, -1, 1))) + 50 * (1 - abs(m[9:] @ tdir)) for m in cand) < 1.5
    assert abs(np.linalg.norm(h["C"]) - 1) < 1e-9
    n_geo, n_cons, n_pose, ok = o[pos:pos + 4]; pos += 4
    assert abs(n_geo - len(h["inliers"])) <= 0.03 * len(h["inliers"]) and n_cons == n_geo and n_pose == 1 and ok == 1.0
    assert o[pos] == 1.0 and o[pos + 1] < 10; pos += 2
    # decomposeHomography on the exact (scaled) homography: four motions, two rotations each with +-t; one pair is the scene's
    assert o[pos] == 0.0 and o[pos + 1] == 4.0; pos += 2
    mot = o[pos:pos + 48].reshape(4, 12); pos += 48
    best = 1e9
    for m in mot:
        R, c = m[:9].reshape(3, 3), m[9:]
        assert abs(np.linalg.det(R) - 1) < 1e-9 and np.abs(R @ R.T - np.eye(3)).max() < 1e-9 and abs(np.linalg.norm(c) - 1) < 1e-12
        best = min(best, np.abs(R - pla["R"]).max() + np.abs(c - tdir).max())
    assert best < 1e-8
    assert np.array_equal(mot[0][:9], mot[1][:9]) and np.array_equal(mot[0][9:], -mot[1][9:])
    assert np.array_equal(mot[2][:9], mot[3][:9]) and np.array_equal(mot[2][9:], -mot[3][9:])
    # a pure rotation: one motion, the rotation itself, zero translation
    assert o[pos] == 1.0
    assert np.abs(o[pos + 1:pos + 10].reshape(3, 3) - Rz).max() < 1e-9 and not o[pos + 10:pos + 13].any()
