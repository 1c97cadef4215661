"""The four HIP policy classes chained the way ColoC chains the reference's (coloc.hpp:150-223: detectFeaturesFile ->
computeMatches -> filterMatches -> setMapData / matchSceneWithMap -> localizeImage), in C++, on rendered frames of one
scene: the relative pose of the camera pair and the absolute pose of the second camera against a map made of the first
camera's features must be the poses the frames were rendered from."""
import os
import subprocess

import numpy as np
import pytest

import synth
from test_policy_host import build_driver

pytestmark = pytest.mark.gpu

W, H = 640, 480
K = np.array([[520.0, 0, 320.0], [0, 520.0, 240.0], [0, 0, 1.0]])
PPU = 100.0


def _pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n# rendered\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def test_cpp_pipeline_on_rendered_frames(tmp_path):
    exe = build_driver(str(tmp_path / "pipeline_driver"), "pipeline_driver.cpp")
    # a textured surface with gentle relief: on an exact plane the essential matrix has two valid decompositions (the planar
    # ambiguity) and either may win the chirality vote
    tex = synth.plane_texture()
    relief = synth.smooth_relief()
    Ra, ta = synth.look_at_plane_pose((7.0, 7.0), 5.0, yaw=0.0, tilt=(0.10, -0.06))
    Rb, tb = synth.look_at_plane_pose((7.6, 6.7), 5.2, yaw=0.12, tilt=(-0.08, 0.09))
    _pgm(tmp_path / "cam0.pgm", synth.render_plane(tex, PPU, K, Ra, ta, W, H, relief=relief))
    _pgm(tmp_path / "cam1.pgm", synth.render_plane(tex, PPU, K, Rb, tb, W, H, relief=relief))
    args = [str(tmp_path), str(W), str(H), str(K[0, 0]), str(K[0, 2]), str(K[1, 2])]
    r = subprocess.run([exe, "features"] + args, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    feat0 = np.fromfile(tmp_path / "feat0.bin", dtype=np.float64).reshape(-1, 2)
    assert len(feat0) > 500
    synth.backproject_to_plane(feat0, K, Ra, ta, relief=relief).astype(np.float64).tofile(tmp_path / "map_xyz.bin")
    r = subprocess.run([exe, "run"] + args, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    o = np.fromfile(tmp_path / "pipeline_out.bin", dtype=np.float64)
    n0, n1, n_put, n_geo = (int(v) for v in o[:4])
    Rrel, Crel = o[4:13].reshape(3, 3), o[13:16]
    n_map, status, n_inl = int(o[16]), o[17], int(o[18])
    Rabs, Cabs, rmse = o[19:28].reshape(3, 3), o[28:31], o[31]
    assert n0 == len(feat0) and n1 > 500 and n_put > 100 and n_geo > 0.6 * n_put
    # relative pose of the pair (camera 0 -> camera 1): rotation and baseline DIRECTION (the scale of E is free)
    R_true = Rb @ Ra.T
    t_true = tb - R_true @ ta
    C_true = -R_true.T @ t_true
    ang = np.degrees(np.arccos(np.clip((np.trace(Rrel @ R_true.T) - 1) / 2, -1, 1)))
    cosb = (Crel @ C_true) / (np.linalg.norm(Crel) * np.linalg.norm(C_true))
    print("relative pose: rotation error %.3f deg, baseline cos %.5f; %d / %d putative kept" % (ang, cosb, n_geo, n_put))
    assert ang < 2.5 and cosb > 0.985, (ang, cosb)          # two-view geometry over a shallow relief: looser than the absolute pose below
    # absolute pose of camera 1 against the map built from camera 0
    assert status == 0.0 and n_map > 100 and n_inl > 0.6 * n_map      # false = success
    ang2 = np.degrees(np.arccos(np.clip((np.trace(Rabs @ Rb.T) - 1) / 2, -1, 1)))
    print("absolute pose: rotation error %.3f deg, centre error %.4f, %d / %d map matches inliers" % (ang2, np.linalg.norm(Cabs - (-Rb.T @ tb)), n_inl, n_map))
    assert ang2 < 0.5 and np.linalg.norm(Cabs - (-Rb.T @ tb)) < 0.02 * 5.0 and 0.0 < rmse < 3.0
    # matchMaps (RobustMatcher.hpp:241-370 as called at coloc.hpp:326): every map-to-map match is kept, in order, status false =
    # success, and guidedmatches2.txt holds "f1^T F f2, xL, yL, xR, yR" per match for F = Kinv^T R^T K^T [K R d / |d|]_x
    n_common, mm_status, kept = int(o[32]), o[33], o[34]
    assert n_common > 100 and mm_status == 0.0 and kept == 1.0
    rows = np.loadtxt(tmp_path / "guidedmatches2.txt", delimiter=",").reshape(-1, 5)
    assert len(rows) == n_common
    d = Cabs / np.linalg.norm(Cabs)
    A = K @ Rabs @ d
    Cx = np.array([[0, -A[2], A[1]], [A[2], 0, -A[0]], [-A[1], A[0], 0]])
    F = np.linalg.inv(K).T @ Rabs.T @ K.T @ Cx
    f1 = np.c_[rows[:, 1:3], np.ones(len(rows))]
    f2 = np.c_[rows[:, 3:5], np.ones(len(rows))]
    want = np.einsum("ni,ij,nj->n", f1, F, f2)
    assert np.allclose(rows[:, 0], want, rtol=2e-3, atol=1e-3 * np.abs(want).max())     # the file carries 6 significant digits
