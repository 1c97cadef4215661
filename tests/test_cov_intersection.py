"""Host-side covariance intersection (coloc_amd/host/HIPCovIntersection.hpp; reference
include/coloc/CovIntersection.hpp:24-49) against a numpy/scipy restatement of the same formulas.
Tolerance: omega within 1e-3 (the reference's own search tolerance, :61), fused values accordingly."""
import os
import subprocess

import numpy as np
from scipy.optimize import minimize_scalar

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fused(CA, CB, x):
    return np.linalg.inv(np.linalg.inv(CA) + np.linalg.inv(CB) - np.linalg.inv(x * CA + (1 - x) * CB))


def _spd(rng, scale):
    M = rng.normal(size=(3, 3))
    return M @ M.T * scale + np.eye(3) * 0.01 * scale


def test_cov_intersection_matches_restatement(tmp_path):
    exe = tmp_path / "ci"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-g", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "coloc_amd", "host"),
                           os.path.join(ROOT, "tests", "host", "ci_driver.cpp"), "-o", str(exe)])
    rng = np.random.default_rng(0)
    cases = []
    for k in range(50):
        CA, CB = _spd(rng, 10 ** rng.uniform(-3, 0)), _spd(rng, 10 ** rng.uniform(-3, 0))
        cases.append((CA, CB, rng.normal(size=3), rng.normal(size=3)))
    cases.append((np.eye(3) * 0.01, np.eye(3) * 0.04, np.zeros(3), np.ones(3)))       # isotropic: any omega gives the same trace
    inp = "\n".join(" ".join("%.17g" % v for v in np.concatenate([c[0].ravel(), c[1].ravel(), c[2], c[3]])) for c in cases)
    out = subprocess.run([str(exe)], input=inp, capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == len(cases)
    for (CA, CB, ca, cb), line in zip(cases, out):
        v = np.array([float(t) for t in line.split()])
        x, val, C, p = v[0], v[1], v[2:11].reshape(3, 3), v[11:14]
        f = lambda w: np.trace(_fused(CA, CB, w))
        ref = minimize_scalar(f, bounds=(0, 1), method="bounded", options={"xatol": 1e-6})
        best = min(ref.fun, f(0.0), f(1.0))
        assert 0.0 <= x <= 1.0
        assert val <= best * (1 + 1e-4) + 1e-12                    # as good a minimum as scipy's
        assert np.isclose(val, f(x), rtol=1e-10)
        assert np.allclose(C, _fused(CA, CB, x), rtol=1e-8, atol=1e-14)
        M = np.linalg.inv(x * CA + (1 - x) * CB)
        Kg = C @ (np.linalg.inv(CA) - x * M); Lg = C @ (np.linalg.inv(CB) - (1 - x) * M)
        assert np.allclose(p, Kg @ ca + Lg @ cb, rtol=1e-8, atol=1e-12)
        assert np.all(np.linalg.eigvalsh((C + C.T) / 2) > 0)


def test_c_abi_export_matches_header():
    """clc_cov_intersection (host-only entry point of the .so; works without a GPU) == the same formulas."""
    from coloc_amd import cov_intersection
    rng = np.random.default_rng(5)
    for _ in range(10):
        CA, CB = _spd(rng, 0.05), _spd(rng, 0.02)
        ca, cb = rng.normal(size=3), rng.normal(size=3)
        om, C, p = cov_intersection(CA, CB, ca, cb)
        assert np.allclose(C, _fused(CA, CB, om), rtol=1e-8)
        f = lambda w: np.trace(_fused(CA, CB, w))
        best = min(minimize_scalar(f, bounds=(0, 1), method="bounded").fun, f(0.0), f(1.0))
        assert np.trace(C) <= best * (1 + 1e-4)
