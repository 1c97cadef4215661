"""ctypes access to the host build of the sequential five-point solver (tests/host/fivept_host_lib.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        out = os.path.join(ROOT, "tests", "host", "libfivept_host.so")
        src = os.path.join(ROOT, "tests", "host", "fivept_host_lib.cpp")
        hdr = os.path.join(ROOT, "coloc_amd", "csrc", "fivept.h")
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out])
        _LIB = C.CDLL(out)
        _LIB.fpt_host_solve.restype = C.c_int
    return _LIB


def solve(q1, q2):
    """q1, q2: 5 x 2 normalised coordinates.  Returns the list of 3 x 3 essential matrices."""
    q1 = np.ascontiguousarray(q1, dtype=np.float64).reshape(5, 2)
    q2 = np.ascontiguousarray(q2, dtype=np.float64).reshape(5, 2)
    E = np.zeros(90)
    n = lib().fpt_host_solve(q1.ctypes.data_as(C.c_void_p), q2.ctypes.data_as(C.c_void_p), E.ctypes.data_as(C.c_void_p))
    return [E[9 * k:9 * k + 9].reshape(3, 3).copy() for k in range(n)]


def random_two_view(rng):
    """A random relative pose and 5 points in front of both cameras: (q1, q2, E_true)."""
    ax = 0.3 * rng.uniform(-1, 1, 3)
    t = rng.uniform(-1, 1, 3) * np.array([1.0, 1.0, 0.3])
    th = np.linalg.norm(ax) + 1e-12
    k = ax / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)
    X = np.c_[rng.uniform(-2, 2, (5, 2)), rng.uniform(2, 6, 5)]
    Y = X @ R.T + t
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    return X[:, :2] / X[:, 2:3], Y[:, :2] / Y[:, 2:3], tx @ R
