"""ctypes access to the host build of the product's seven-point / four-point statements (tests/host/twoview_host_lib.cpp over
coloc_amd/csrc/twoview_min.h) and the synthetic two-view scenes the 'F' / 'H' tests share."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        out = os.path.join(ROOT, "tests", "host", "libtwoview_host.so")
        src = os.path.join(ROOT, "tests", "host", "twoview_host_lib.cpp")
        hdr = os.path.join(ROOT, "coloc_amd", "csrc", "twoview_min.h")
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out])
        _LIB = C.CDLL(out)
        _LIB.tv_host_seven_point.restype = C.c_int
        _LIB.tv_host_four_point.restype = C.c_int
        _LIB.tv_host_cubic.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def seven_point(q1, q2):
    q1 = np.ascontiguousarray(q1, dtype=np.float64).reshape(7, 2); q2 = np.ascontiguousarray(q2, dtype=np.float64).reshape(7, 2)
    F = np.zeros(27)
    n = lib().tv_host_seven_point(_p(q1), _p(q2), _p(F))
    return [F[9 * k:9 * k + 9].copy() for k in range(n)]


def four_point(q1, q2):
    q1 = np.ascontiguousarray(q1, dtype=np.float64).reshape(4, 2); q2 = np.ascontiguousarray(q2, dtype=np.float64).reshape(4, 2)
    H = np.zeros(9)
    lib().tv_host_four_point(_p(q1), _p(q2), _p(H))
    return H


def normalizer(wh):
    t = np.zeros(3)
    lib().tv_host_normalizer(C.c_int(int(wh[0])), C.c_int(int(wh[1])), _p(t))
    return t


def unnormalize(homography, wh, Mn):
    Mn = np.ascontiguousarray(Mn, dtype=np.float64).reshape(9)
    out = np.zeros(9)
    lib().tv_host_unnormalize(C.c_int(1 if homography else 0), C.c_int(int(wh[0])), C.c_int(int(wh[1])), _p(Mn), _p(out))
    return out.reshape(3, 3)


def cubic(a, b, c):
    x = np.zeros(3)
    n = lib().tv_host_cubic(C.c_double(a), C.c_double(b), C.c_double(c), _p(x))
    return x[:n]


def unit(M):
    """unit Frobenius norm, the entry of largest magnitude positive"""
    M = np.asarray(M, dtype=np.float64).reshape(9)
    M = M / np.linalg.norm(M)
    return M if M[np.argmax(np.abs(M))] > 0 else -M


K_DEFAULT = np.array([[700.0, 0.0, 640.0], [0.0, 700.0, 360.0], [0.0, 0.0, 1.0]])


def scene(n, seed, planar=False, outlier_frac=0.3, noise=0.3, wh=(1280, 720), K=None):
    """Two views of n points (a plane when `planar`): pixels x1, x2 (n, 2), the true F and H (None unless planar), the relative pose
    (R, t: X2 = R X1 + t), the plane (normal, distance) and the indices turned into outliers."""
    rng = np.random.default_rng(seed)
    K = K_DEFAULT if K is None else K
    ax = 0.25 * rng.uniform(-1, 1, 3)
    th = np.linalg.norm(ax) + 1e-12
    k = ax / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
    t = rng.uniform(-1, 1, 3) * np.array([0.8, 0.5, 0.2])
    # points that project inside image 1
    u = np.c_[rng.uniform(40, wh[0] - 40, n), rng.uniform(40, wh[1] - 40, n)]
    rays = np.c_[u, np.ones(n)] @ np.linalg.inv(K).T
    normal = np.array([0.15, -0.1, 1.0]); normal /= np.linalg.norm(normal)
    dist = 5.0
    if planar:
        depth = dist / (rays @ normal)
    else:
        depth = rng.uniform(3.0, 9.0, n)
    X = rays * depth[:, None]
    Y = X @ R.T + t
    x1 = (X / X[:, 2:]) @ K.T
    x2 = (Y / Y[:, 2:]) @ K.T
    x1 = x1[:, :2] + noise * rng.standard_normal((n, 2))
    x2 = x2[:, :2] + noise * rng.standard_normal((n, 2))
    n_out = int(outlier_frac * n)
    out = rng.choice(n, n_out, replace=False)
    x2[out] = np.c_[rng.uniform(0, wh[0], n_out), rng.uniform(0, wh[1], n_out)]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(K)
    F = Ki.T @ tx @ R @ Ki
    H = K @ (R + np.outer(t, normal) / dist) @ Ki if planar else None
    return dict(x1=np.ascontiguousarray(x1), x2=np.ascontiguousarray(x2), F=F, H=H, R=R, t=t, normal=normal, dist=dist, outliers=out, K=K, wh=wh)
