"""Host-side pose filter and pose CSV (coloc_amd/host/HIPPoseFilter.hpp, HIPPoseLog.hpp; reference
include/coloc/KalmanFilter.hpp:8-165, logUtils.hpp:36-100) against an independent numpy restatement of the
same equations.  Tolerance 1e-9 relative on the fp64 state (different but equivalent solves)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = tmp_path_factory.mktemp("pf") / "pf"
    # ASan + UBSan: these headers are host code, the only place where sanitizers can run (none on the GPU pool)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-g", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "coloc_amd", "host"),
                           os.path.join(ROOT, "tests", "host", "pose_filter_driver.cpp"), "-o", str(exe)])
    return str(exe)


def _run(exe, text):
    return subprocess.run([exe], input=text, capture_output=True, text=True, check=True).stdout.splitlines()


def _fmt(a):
    return " ".join("%.17g" % v for v in np.asarray(a, dtype=np.float64).ravel())


def _euler2rot(e):            # colocUtils.hpp:102-141
    b, a, h = e
    ch, sh, ca, sa, cb, sb = np.cos(h), np.sin(h), np.cos(a), np.sin(a), np.cos(b), np.sin(b)
    return np.array([[ch * ca, sh * sb - ch * sa * cb, ch * sa * sb + sh * cb],
                     [sa, ca * cb, -ca * sb],
                     [-sh * ca, sh * sa * cb + ch * sb, -sh * sa * sb + ch * cb]])


def _rot2euler(R):            # colocUtils.hpp:63-100
    if R[1, 0] > 0.998:
        return np.array([0.0, np.pi / 2, np.arctan2(R[0, 2], R[2, 2])])
    if R[1, 0] < -0.998:
        return np.array([0.0, -np.pi / 2, np.arctan2(R[0, 2], R[2, 2])])
    return np.array([np.arctan2(-R[1, 2], R[1, 1]), np.arcsin(R[1, 0]), np.arctan2(-R[2, 0], R[0, 0])])


class _RefFilter:
    """cv::KalmanFilter semantics with A = H = I, restated in numpy (KalmanFilter.hpp:115-128)."""
    def __init__(self, n):
        self.x = [np.zeros(6) for _ in range(n)]
        self.P = [np.eye(6) for _ in range(n)]
        self.R = [np.eye(6) * 1e-1 for _ in range(n)]
        self.z = [np.zeros(6) for _ in range(n)]
        self.avail = False
        self.init = True

    def fill(self, d, t, Rm):
        self.z[d] = np.concatenate([t, _rot2euler(Rm)])
        self.avail = True

    def update(self, d, cov, rmse):
        xp = self.x[d].copy()
        Pp = self.P[d] + np.eye(6) * 1e-2
        self.P[d] = Pp.copy()
        c = np.asarray(cov).reshape(6, 6)
        self.R[d][3:, 3:] = c[3:, 3:] * np.float64(np.float32(rmse))
        est, gate, rejected = xp, None, False
        if self.avail:
            S = Pp + self.R[d]
            y = self.z[d] - xp
            gate = y @ S @ y                      # the reference multiplies by S, not its inverse (:148)
            if gate > 10 and not self.init:
                rejected = True
            else:
                K = Pp @ np.linalg.inv(S)
                est = xp + K @ y
                self.P[d] = Pp - K @ Pp
        self.x[d] = est.copy()
        self.avail = False
        if d == 2:
            self.init = False
        return _euler2rot(est[3:]), est[:3], gate, rejected, self.init, self.P[d]


def _rand_rot(rng, small=False):
    from scipy.spatial.transform import Rotation
    if small:
        return Rotation.from_rotvec(rng.normal(size=3) * 0.2).as_matrix()
    return Rotation.random(random_state=int(rng.integers(1 << 30))).as_matrix()


def test_filter_sequence_matches_restatement(driver):
    rng = np.random.default_rng(11)
    n = 3
    ref = _RefFilter(n)
    script, expect = ["F %d" % n], []
    for step in range(40):
        d = step % n
        cov = np.zeros((6, 6))
        M = rng.normal(size=(3, 3)); cov[3:, 3:] = (M @ M.T) * 1e-3
        M = rng.normal(size=(3, 3)); cov[:3, :3] = (M @ M.T) * 1e-4
        rmse = float(np.float32(rng.uniform(0.3, 2.0)))
        if step % 7 != 5:                          # every seventh update has no measurement: prediction only
            t = rng.normal(size=3) * (0.05 if step % 11 else 5.0) + np.array([1.0, 2.0, 3.0]) * (d + 1)   # the occasional outlier trips the gate
            Rm = _rand_rot(rng, small=True)
            ref.fill(d, t, Rm)
            script.append("M %d %s %s" % (d, _fmt(t), _fmt(Rm)))
        expect.append(ref.update(d, cov.ravel(), rmse))
        script.append("U %d %.9g %s" % (d, rmse, _fmt(cov)))
    out = _run(driver, "\n".join(script) + "\n")
    assert len(out) == len(expect)
    n_rejected = 0
    for line, (Rm, t, gate, rejected, init, P) in zip(out, expect):
        v = np.array([float(x) for x in line.split()])
        assert np.allclose(v[:9].reshape(3, 3), Rm, rtol=1e-9, atol=1e-12)
        assert np.allclose(v[9:12], t, rtol=1e-9, atol=1e-12)
        if gate is not None:
            assert np.isclose(v[12], gate, rtol=1e-9)
        assert int(v[13]) == int(rejected) and int(v[14]) == int(init)
        assert np.allclose(v[15:].reshape(6, 6), P, rtol=1e-9, atol=1e-14)
        n_rejected += int(rejected)
    assert n_rejected >= 1                          # the gate did fire after the initial phase
    assert not expect[-1][4]


def test_euler_angles_reconstruct_rotation_and_pole_cut(driver):
    rng = np.random.default_rng(3)
    Rs = [_rand_rot(rng) for _ in range(200)]
    out = _run(driver, "\n".join("E " + _fmt(R) for R in Rs) + "\n")
    for R, line in zip(Rs, out):
        e = np.array([float(x) for x in line.split()])
        a0, a1, a2 = e[:3]
        assert 0.0 <= a0 <= np.pi + 1e-12           # Eigen's range for the first angle
        Rz = np.array([[np.cos(a0), -np.sin(a0), 0], [np.sin(a0), np.cos(a0), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(a1), 0, np.sin(a1)], [0, 1, 0], [-np.sin(a1), 0, np.cos(a1)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(a2), -np.sin(a2)], [0, np.sin(a2), np.cos(a2)]])
        assert np.allclose(Rz @ Ry @ Rx, R, atol=1e-12)
        # logUtils.hpp:36-67 restated
        b1, b2, b3 = np.float32(a0 * 180 / np.pi), np.float32(a2 * 180 / np.pi), np.float32(a1 * 180 / np.pi)
        if abs(b2) > 120:
            b2 = (-b2 - 180) if b2 < 0 else 180 - b2
        if abs(b3) > 120:
            b3 = 180 + b3 if b3 < 0 else b3 - 180
        else:
            b3 = -b3
        if abs(b1) > 120:
            b1 = 180 + b1 if b1 < 0 else b1 - 180
        assert np.allclose(e[3:], np.array([b1, b2, b3], dtype=np.float64) * np.pi / 180, rtol=1e-6, atol=1e-7)
    # rot2euler / euler2rot round trip away from the poles (exercised through the filter's measurement path above)
    for _ in range(50):
        e = rng.uniform(-1.2, 1.2, size=3)
        assert np.allclose(_rot2euler(_euler2rot(e)), e, atol=1e-12)


def test_pose_csv_record_format(driver):
    rng = np.random.default_rng(5)
    R = _rand_rot(rng)
    c = np.array([1.25, -0.000123456789, 12345.678])
    cov = np.arange(36, dtype=np.float64) * 1e-3 + 1e-7
    line = _run(driver, "L 17 2 5 %s %s %s 0.75 321\n" % (_fmt(R), _fmt(c), _fmt(cov)))[0]
    f = line.split(",")
    assert len(f) == 20
    assert f[0] == "17" and f[1] == "5" and f[2] == "2"          # idx, DEST, SOURCE (logUtils.hpp:94)
    assert f[3:6] == ["%g" % v for v in c]                        # default ostream formatting: 6 significant digits
    assert f[6:15] == ["%g" % cov[i] for i in (21, 22, 23, 27, 28, 29, 33, 34, 35)]
    assert f[18] == "0.75" and f[19] == "321"
    e = np.array([float(x) for x in _run(driver, "E %s\n" % _fmt(R))[0].split()])[3:]
    assert [float(x) for x in f[15:18]] == pytest.approx(list(e * 180 / np.pi), rel=1e-5, abs=1e-4)


def test_ply_dumps(driver, tmp_path):
    """logMaptoPLY / logPosetoPLY (logUtils.hpp:102-167): ASCII PLY, green vertices for the posed views, white ones for the landmarks,
    fixed notation with digits10 + 1 = 16 decimals; the track file gets one appended vertex line per call and no header."""
    rng = np.random.default_rng(3)
    poses, pts = rng.normal(size=(2, 3)), rng.normal(size=(5, 3)) * 10
    path = tmp_path / "map.ply"
    out = _run(driver, "P %s 2 5 %s\n" % (path, " ".join(repr(float(v)) for v in np.r_[poses.ravel(), pts.ravel()])))
    assert out == ["1"]
    lines = open(path).read().split("\n")
    assert lines[:10] == ["ply", "format ascii 1.0", "comment generated by coloc", "element vertex 7", "property double x", "property double y",
                          "property double z", "property uchar red", "property uchar green", "property uchar blue"]
    assert lines[10] == "end_header" and len(lines) == 11 + 7 + 1 and lines[-1] == ""
    body = lines[11:18]
    for k, row in enumerate(body):
        f = row.split(" ")
        want = poses[k] if k < 2 else pts[k - 2]
        assert f[3:] == (["0", "255", "0"] if k < 2 else ["255", "255", "255"])
        assert all(len(v.split(".")[1]) == 16 for v in f[:3]) and np.allclose([float(v) for v in f[:3]], want, rtol=0, atol=1e-15)
    track = open(str(path) + ".track").read().split("\n")
    assert len(track) == 3 and track[2] == "" and all(t.endswith(" 0 255 0") for t in track[:2])
