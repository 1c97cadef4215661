"""CPU tests of the a-contrario RANSAC oracle (oracle/clc_oracle_acr.c) and of the PRODUCT's own statement of the same
arithmetic (coloc_amd/csrc/clc_acr.h, reached through its host build tests/host/acr_host_lib.cpp).  Since round 6 the two
share no code: the product's portable log10 is held against libm (which is what the oracle evaluates), the product's
sampler -- run-time and compile-time-size forms -- against the oracle's independent statement of the documented sampler,
the product's NFA term against the oracle's; then the oracle's log-combination tables, its NFA scan vs a numpy
restatement, and the whole loop vs an independent pure-Python restatement of the published algorithm
(Moisan-Moulon-Monasse, IPOL 2012; OpenMVG's ACRANSAC as called at reference Localizer.hpp:82-93)."""
import math

import numpy as np
import pytest
from scipy.special import gammaln

import acr_host
import synth
from solvers_np import numpy_p3p

FLT_EPS = float(np.finfo(np.float32).eps)


def test_products_portable_log10_within_two_ulp_of_libm():
    """The GPU evaluates clc_acr_log10 (host and device give the same bits: IEEE +, -, *, / only); the oracle evaluates libm's."""
    rng = np.random.default_rng(5)
    xs = np.concatenate([10.0 ** rng.uniform(-12, 12, 200000), rng.uniform(0.5, 2.0, 100000), 1.0 + rng.uniform(-1e-6, 1e-6, 20000),
                         np.arange(1, 20001, dtype=np.float64), [FLT_EPS, 1.0, 10.0, 1e-300, 5e-324, 1e308, math.pi]])
    got = np.array([acr_host.log10(v) for v in xs[:60000]])
    want = np.log10(xs[:60000])
    ulp = np.spacing(np.abs(want))
    assert np.all(np.abs(got - want) <= 2.0 * ulp + 1e-300) and (np.abs(got - want) > 0).mean() < 0.03
    for v in xs[-7:]:
        assert abs(acr_host.log10(v) - math.log10(v)) <= 2 * np.spacing(abs(math.log10(v))) + 1e-300
    assert acr_host.log10(1.0) == 0.0 and acr_host.log10(100.0) == 2.0
    # the float log10 table the log-combination tables are summed from: the product's and libm's round to the same float for every k
    # a solve can meet (16 384 correspondences at most)
    ks = np.arange(1, 16386, dtype=np.float64)
    assert np.array_equal(np.array([acr_host.log10(k) for k in ks]).astype(np.float32), np.log10(ks).astype(np.float32))


def test_products_sampler_equals_the_oracles_statement_of_it(oracle):
    """The sample of an iteration is a documented pure function of (seed, iteration, index-set size) -- the product states it in
    clc_acr.h (a run-time-size form on the host, a compile-time-size form in the kernels), the oracle states it again on its own."""
    rng = np.random.default_rng(11)
    cases = [(42, it, n, m) for m in (3, 5) for n in (m + 1, 7, 100, 5000, 16384) for it in (0, 1, 255, 499999)]
    cases += [(int(rng.integers(0, 2 ** 63)), int(rng.integers(0, 500000)), int(rng.integers(6, 16385)), int(rng.choice([3, 5]))) for _ in range(4000)]
    for seed, it, n, m in cases:
        want = oracle.acr_sample(seed, it, n, m)
        assert acr_host.sample(seed, it, n, m) == want and acr_host.sample(seed, it, n, m, fixed=True) == want, (seed, it, n, m)


def test_products_nfa_term_agrees_with_the_oracles(oracle):
    """clc_acr_nfa (portable log10) against the oracle's scan (libm) on the same residuals: the same k, the value to 1e-13 relative."""
    rng = np.random.default_rng(12)
    for n, frac in ((60, 0.6), (700, 0.4)):
        ninl = int(frac * n)
        err = np.sort(np.concatenate([rng.exponential(1e-6, ninl), rng.uniform(1e-3, 0.5, n - ninl)]))
        cn, ck = oracle.acr_tables(n, 3)
        la0, loge0 = math.log10(math.pi), math.log10(4 * (n - 3))
        vals = [acr_host.nfa(loge0, la0, 1.0, float(err[k - 1]), k, 3, float(cn[k]), float(ck[k])) for k in range(4, n + 1)]
        want, wk = oracle.acr_best_nfa(err, 3, 4, la0, 1.0)
        assert 4 + int(np.argmin(vals)) == wk and abs(min(vals) - want) <= 1e-13 * abs(want)


def test_sampler_is_a_pure_function_with_distinct_positions(oracle):
    for m in (3, 5):
        for n_index in (m + 1, 7, 100, 5000):
            for it in (0, 1, 255):
                a = oracle.acr_sample(42, it, n_index, m)
                assert a == oracle.acr_sample(42, it, n_index, m)
                assert len(set(a)) == m and all(0 <= v < n_index for v in a)
    # different iterations / seeds give different samples; roughly uniform positions
    cnt = np.zeros(50)
    for it in range(4000):
        for v in oracle.acr_sample(7, it, 50, 3):
            cnt[v] += 1
    assert cnt.min() > 150 and cnt.max() < 330
    assert oracle.acr_sample(1, 0, 1000, 3) != oracle.acr_sample(2, 0, 1000, 3)


@pytest.mark.parametrize("n,m", [(10, 3), (200, 3), (1000, 5), (5000, 3)])
def test_log_combination_tables(oracle, n, m):
    cn, ck = oracle.acr_tables(n, m)
    k = np.arange(n + 1)
    want_n = (gammaln(n + 1) - gammaln(k + 1) - gammaln(n - k + 1)) / math.log(10)
    want_n[0] = 0; want_n[n] = 0
    assert np.allclose(cn, want_n, rtol=2e-4, atol=2e-3)
    want_k = np.where(k > m, (gammaln(k + 1) - gammaln(m + 1) - gammaln(np.maximum(k - m, 0) + 1)) / math.log(10), 0.0)
    assert np.allclose(ck, want_k, rtol=2e-4, atol=2e-3)


def _np_best_nfa(err, m, M, logalpha0, mult, cn, ck):
    n = len(err)
    order = np.lexsort((np.arange(n), err))
    e = err[order]
    loge0 = math.log10(M * (n - m))
    best, bk = float("inf"), m
    for k in range(m + 1, n + 1):
        la = logalpha0 + mult * math.log10(e[k - 1] + FLT_EPS)
        v = loge0 + la * (k - m) + float(cn[k]) + float(ck[k])
        if v < best:
            best, bk = v, k
    return best, bk, order


def test_nfa_scan_vs_numpy(oracle):
    rng = np.random.default_rng(8)
    for n, frac in ((50, 0.6), (400, 0.3), (1500, 0.8)):
        ninl = int(frac * n)
        err = np.concatenate([rng.exponential(1e-6, ninl), rng.uniform(1e-3, 0.5, n - ninl)])
        rng.shuffle(err)
        cn, ck = oracle.acr_tables(n, 3)
        want, wk, _ = _np_best_nfa(err, 3, 4, math.log10(math.pi), 1.0, cn, ck)
        got, gk = oracle.acr_best_nfa(err, 3, 4, math.log10(math.pi), 1.0)
        assert gk == wk and abs(got - want) < 1e-9 * max(1.0, abs(want))
        assert abs(wk - ninl) <= max(3, 0.05 * n) and want < 0


def _py_acransac(oracle, X, x, K, fit, max_iter, seed):
    """Independent restatement of the loop in plain Python (libm log10, numpy sort)."""
    n, m, M = X.shape[0], 3, 4
    cn, ck = oracle.acr_tables(n, m)
    s = 1.0 / K[0, 0]
    index = list(range(n))
    inliers, min_nfa, model, emax = [], float("inf"), None, float("inf")
    reserve = max_iter // 10
    n_iter = max_iter - reserve
    it = 0
    while it < n_iter:
        pos = oracle.acr_sample(seed, it, len(index), m)
        sample = [index[p] for p in pos]
        better = False
        for Rt in fit(sample):
            Rt = np.asarray(Rt).reshape(3, 4)
            pc = X @ Rt[:, :3].T + Rt[:, 3]
            uvw = pc @ K.T
            r = (x - uvw[:, :2] / uvw[:, 2:3]) * s
            err = (r ** 2).sum(1)
            v, k, order = _np_best_nfa(err, m, M, math.log10(math.pi), 1.0, cn, ck)
            if v < min_nfa:
                better, min_nfa, model, inliers, emax = True, v, Rt.copy(), [int(i) for i in order[:k]], float(err[order[k - 1]])
        if (better and min_nfa < 0) or (it + 1 == n_iter and reserve):
            if not inliers:
                n_iter += 1; reserve -= 1
            else:
                index = list(inliers)
                if reserve:
                    n_iter = it + 1 + reserve; reserve = 0
        it += 1
    if min_nfa >= 0:
        inliers = []
    return model, inliers, min_nfa, (math.sqrt(emax) / s if inliers else 0.0), it


@pytest.mark.parametrize("n,outl,seed", [(60, 0.3, 1), (150, 0.5, 2), (300, 0.2, 3)])
def test_whole_loop_vs_pure_python_restatement(oracle, n, outl, seed):
    sc = synth.pnp_scene(n, seed=4300 + seed, outlier_frac=outl)
    X, x, K = sc["X"], sc["x"], sc["K"]

    def fit(sample):
        return [P.reshape(-1) for P in numpy_p3p(X[sample], x[sample], K)]

    res = oracle.acransac(0, X, x, K, fit, max_iteration=48, seed=seed)
    model, inl, nfa, emax, its = _py_acransac(oracle, X, x, K, fit, 48, seed)
    assert res["found"] and res["iterations"] == its
    assert list(res["inliers"]) == inl
    assert np.allclose(res["model"].reshape(3, 4), model, atol=1e-12)
    assert abs(res["min_nfa"] - nfa) < 1e-9 * abs(nfa) and abs(res["error_max"] - emax) < 1e-9 * emax
    # the a-contrario threshold separates the planted inliers: all but a few true inliers in, (almost) no outlier in
    got = np.zeros(n, bool); got[res["inliers"]] = True
    assert (got & sc["inliers"]).sum() >= 0.9 * sc["inliers"].sum() and (got & ~sc["inliers"]).sum() <= max(1, 0.03 * n)
    # the phase switch: once a meaningful model exists, samples come from its inliers and only the reserve is run
    first = res["samples"][: res["iterations"]]
    assert res["iterations"] <= 48 and all(len(set(s)) == 3 for s in first)


def test_degenerate_inputs(oracle):
    sc = synth.pnp_scene(3, seed=1, outlier_frac=0.0)
    res = oracle.acransac(0, sc["X"], sc["x"], sc["K"], lambda s: [], max_iteration=16)
    assert not res["found"] and res["iterations"] == 0
    # pure noise: no meaningful model (NFA >= 0) -> no inliers; every reserve iteration is spent looking
    rng = np.random.default_rng(3)
    X = rng.uniform(-5, 5, (80, 3)) + [0, 0, 12]
    x = np.stack([rng.uniform(0, 1280, 80), rng.uniform(0, 720, 80)], 1)
    K = np.array([[1000.0, 0, 640], [0, 1000.0, 360], [0, 0, 1]])
    res = oracle.acransac(0, X, x, K, lambda s: [P.reshape(-1) for P in numpy_p3p(X[s], x[s], K)], max_iteration=40, seed=9)
    assert not res["found"] and len(res["inliers"]) == 0 and res["min_nfa"] >= 0 and res["iterations"] == 40
