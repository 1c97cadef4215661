"""ctypes access to the host build of the P3P solver (tests/host/p3p_host_lib.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        out = os.path.join(ROOT, "tests", "host", "libp3p_host.so")
        src = os.path.join(ROOT, "tests", "host", "p3p_host_lib.cpp")
        hdr = os.path.join(ROOT, "coloc_amd", "csrc", "p3p.h")
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", src, "-o", out])
        _LIB = C.CDLL(out)
        _LIB.p3p_host_sample.restype = C.c_int
    return _LIB


def sample_poses(X, x, K, sample):
    """The four pose slots (4 x 3 x 4, NaN where the root has no pose) of the P3P problem on correspondences `sample`."""
    X = np.ascontiguousarray(X, dtype=np.float64); x = np.ascontiguousarray(x, dtype=np.float64)
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    smp = np.ascontiguousarray(sample, dtype=np.int32)
    out = np.zeros(48)
    lib().p3p_host_sample(X.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), K.ctypes.data_as(C.c_void_p),
                          smp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out.reshape(4, 3, 4)
