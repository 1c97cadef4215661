"""ctypes access to the host build of the product's a-contrario arithmetic (tests/host/acr_host_lib.cpp over coloc_amd/csrc/clc_acr.h)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        out = os.path.join(ROOT, "tests", "host", "libacr_host.so")
        src = os.path.join(ROOT, "tests", "host", "acr_host_lib.cpp")
        hdr = os.path.join(ROOT, "coloc_amd", "csrc", "clc_acr.h")
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out])
        _LIB = C.CDLL(out)
        _LIB.acr_host_log10.restype = C.c_double
        _LIB.acr_host_log10.argtypes = [C.c_double]
        _LIB.acr_host_nfa.restype = C.c_double
        _LIB.acr_host_nfa.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_float, C.c_float]
    return _LIB


def log10(x):
    return float(lib().acr_host_log10(float(x)))


def sample(seed, it, n_index, m, fixed=False):
    pos = (C.c_uint32 * 8)()
    fn = lib().acr_host_sample_fixed if fixed else lib().acr_host_sample
    fn(C.c_uint64(int(seed)), C.c_uint32(int(it)), C.c_uint32(int(n_index)), C.c_int(m), pos)
    return [int(pos[j]) for j in range(m)]


def nfa(loge0, logalpha0, mult, e_k, k, m, logc_n_k, logc_k_k):
    return float(lib().acr_host_nfa(loge0, logalpha0, mult, e_k, k, m, logc_n_k, logc_k_k))
