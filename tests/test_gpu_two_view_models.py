"""The a-contrario filter under RobustMatcher's 'F' and 'H' models on the GPU (clc_two_view_acransac, coloc_amd/csrc/acransac.hip +
twoview_min.h; reference include/coloc/RobustMatcher.hpp:128-151, :188-239, dispatch :399-405) against the sequential oracle
(oracle/clc_oracle_acr.c kinds 2 / 3, oracle/clc_oracle_twoview.c).  As for the other two kinds (test_gpu_acransac.py) in two ways:
the oracle fed with the DEVICE's minimal models (clc_two_view_minimal) must be reproduced exactly -- model, inlier list in order,
threshold, iteration count, NFA to NFA_RTOL --, which checks conditioning, residuals, ordering, NFA, model selection and the phase
switch; the oracle fed with its OWN solvers (one-sided Jacobi SVD; nothing from the GPU) must find the same solution of the scene,
and the same numbers to rounding where both runs pick the model of the same iteration."""
import numpy as np
import pytest

import twoview_host as tvh
from coloc_amd import abi
from test_gpu_acransac import NFA_RTOL, _same_nfa

pytestmark = pytest.mark.gpu
KIND = {"F": 2, "H": 3}


def _device_fit(ctx, model, sc):
    def fit(sample):
        mo = ctx.two_view_minimal(model, sc["x1"], sc["x2"], sc["wh"], np.array([sample], dtype=np.int32))[0]
        return [m for m in mo if not np.isnan(m).any()]
    return fit


def _oracle_fit(oracle, model, sc):
    q1, q2 = oracle.tv_normalize(sc["wh"], sc["x1"]), oracle.tv_normalize(sc["wh"], sc["x2"])

    def fit(sample):
        return oracle.seven_point(q1[sample], q2[sample]) if model == "F" else [oracle.four_point(q1[sample], q2[sample])]
    return fit


def _check_exact(ctx, oracle, model, sc, max_it=256, seed=1, precision=float("inf")):
    got = ctx.two_view_acransac(model, sc["x1"], sc["x2"], sc["wh"], max_iteration=max_it, seed=seed, precision=precision)
    want = oracle.acransac(KIND[model], sc["x1"], sc["x2"], np.eye(3), _device_fit(ctx, model, sc), max_iteration=max_it, seed=seed,
                           precision=precision, img_wh=sc["wh"])
    assert (got["M"] is not None) == want["found"]
    assert got["iterations"] == want["iterations"]
    assert _same_nfa(got["min_nfa"], want["min_nfa"])
    assert np.array_equal(got["inliers"], want["inliers"].astype(np.int32))
    if want["found"]:
        assert np.array_equal(got["M"].reshape(-1), want["model"])
        assert got["error_max"] == want["error_max"]
        m = np.zeros(len(sc["x1"]), bool); m[want["inliers"]] = True
        assert np.array_equal(got["mask"], m)
        if model == "F":
            assert np.array_equal(got["F"], got["M"])
        else:
            assert not got["F"].any()
    else:
        assert not got["mask"].any()
    return got, want


@pytest.mark.parametrize("model", ["F", "H"])
@pytest.mark.parametrize("n,seed", [(300, 41), (1000, 42), (1024, 43), (2500, 44), (5000, 45)])
def test_equals_sequential_oracle(gpu_ctx, oracle, model, n, seed):
    sc = tvh.scene(n, seed, planar=model == "H")
    got, want = _check_exact(gpu_ctx, oracle, model, sc, seed=seed)
    assert want["found"]
    true_in = np.ones(n, bool); true_in[sc["outliers"]] = False
    assert (got["mask"] & true_in).sum() >= 0.9 * true_in.sum()
    if model == "F":
        assert np.abs(tvh.unit(got["M"]) - tvh.unit(sc["F"])).max() < 0.05
    else:
        y = np.c_[sc["x1"], np.ones(n)] @ got["M"].T
        assert np.median(np.linalg.norm(sc["x2"] - y[:, :2] / y[:, 2:], axis=1)[true_in]) < 1.5
    assert 0.3 < got["error_max"] < 6.0


@pytest.mark.parametrize("model", ["F", "H"])
def test_more_than_8192_correspondences(gpu_ctx, oracle, model):
    """16 elements per thread in the sort (the widest form of the round kernel), and the capacity is reported past 16 384"""
    sc = tvh.scene(9000, 46, planar=model == "H")
    got, want = _check_exact(gpu_ctx, oracle, model, sc, max_it=40, seed=46)
    assert want["found"] and len(got["inliers"]) > 5000
    big = tvh.scene(16385, 47, planar=model == "H")
    with pytest.raises(abi.CLCError):
        gpu_ctx.two_view_acransac(model, big["x1"], big["x2"], big["wh"], max_iteration=8)


@pytest.mark.parametrize("model", ["F", "H"])
def test_other_outlier_rates_iteration_counts_and_seeds(gpu_ctx, oracle, model):
    for n, outl, max_it, seed in [(200, 0.0, 64, 1), (400, 0.5, 256, 2), (800, 0.6, 512, 3), (150, 0.2, 16, 4), (64, 0.3, 40, 5)]:
        sc = tvh.scene(n, 50 + seed, planar=model == "H", outlier_frac=outl)
        _check_exact(gpu_ctx, oracle, model, sc, max_it=max_it, seed=seed)


@pytest.mark.parametrize("model", ["F", "H"])
def test_upper_bound_mode(gpu_ctx, oracle, model):
    """a finite precision (RelativePose_Info::initial_residual_tolerance, RobustMatcher.hpp:141): residuals above precision x N2(0,0)^2
    never count and the a-contrario mode only starts once a model has more than 2.5 m of them under the bound"""
    sc = tvh.scene(600, 61, planar=model == "H")
    for precision in (4.0, 1.0, 0.04):
        got, want = _check_exact(gpu_ctx, oracle, model, sc, seed=7, precision=precision)
        if want["found"]:
            assert got["error_max"] <= np.sqrt(precision) * (1 + 1e-9)


@pytest.mark.parametrize("model", ["F", "H"])
def test_no_model_cases(gpu_ctx, oracle, model):
    rng = np.random.default_rng(3)
    m = 7 if model == "F" else 4
    sc = dict(x1=np.c_[rng.uniform(0, 1280, 80), rng.uniform(0, 720, 80)], x2=np.c_[rng.uniform(0, 1280, 80), rng.uniform(0, 720, 80)], wh=(1280, 720))
    got, want = _check_exact(gpu_ctx, oracle, model, sc, max_it=64, seed=3)
    if model == "H":
        assert got["M"] is None and len(got["inliers"]) == 0                    # random points share no homography
    # not more data than a sample / no iterations: nothing is run
    for n in (0, m - 1, m):
        r = gpu_ctx.two_view_acransac(model, sc["x1"][:n], sc["x2"][:n], sc["wh"], max_iteration=64, seed=1)
        assert r["M"] is None and r["iterations"] == 0 and np.isinf(r["min_nfa"])
    r = gpu_ctx.two_view_acransac(model, sc["x1"], sc["x2"], sc["wh"], max_iteration=0)
    assert r["M"] is None and r["iterations"] == 0
    with pytest.raises(abi.CLCError):
        gpu_ctx.two_view_acransac(model, sc["x1"], sc["x2"], (0, 720))
    with pytest.raises(abi.CLCError):
        gpu_ctx.two_view_acransac("Q", sc["x1"], sc["x2"], sc["wh"])


@pytest.mark.parametrize("model", ["F", "H"])
def test_minimal_models_device_against_host_build_and_oracle(gpu_ctx, oracle, model):
    """the device's seven-point / four-point models of given samples: the same source built for the host gives the same matrices to the
    last bits but the transcendental functions in the cubic (acos, cos, cbrt: device library vs libm), the oracle's own solvers (another
    null-space basis) the same matrices to rounding"""
    sc = tvh.scene(500, 71, planar=model == "H", outlier_frac=0.0, noise=0.2)
    rng = np.random.default_rng(8)
    m = 7 if model == "F" else 4
    samples = np.array([rng.choice(500, m, replace=False) for _ in range(200)], dtype=np.int32)
    dev = gpu_ctx.two_view_minimal(model, sc["x1"], sc["x2"], sc["wh"], samples)
    q1, q2 = oracle.tv_normalize(sc["wh"], sc["x1"]), oracle.tv_normalize(sc["wh"], sc["x2"])
    worst_host = worst_orc = 0.0
    for s, mo in zip(samples, dev):
        mo = [x for x in mo if not np.isnan(x).any()]
        host = tvh.seven_point(q1[s], q2[s]) if model == "F" else [tvh.four_point(q1[s], q2[s])]
        orc = oracle.seven_point(q1[s], q2[s]) if model == "F" else [oracle.four_point(q1[s], q2[s])]
        assert len(mo) == len(host)
        for a, b in zip(mo, host):
            worst_host = max(worst_host, float(np.abs(a - b).max() / np.abs(b).max()))
        if len(orc) == len(mo):
            for a in mo:
                worst_orc = max(worst_orc, min(float(np.abs(tvh.unit(a) - tvh.unit(b)).max()) for b in orc))
    print("%s: device vs host build %.1e (relative), device vs oracle's solver %.1e (unit norm)" % (model, worst_host, worst_orc))
    assert worst_host < 1e-9 and worst_orc < 1e-6
    # the device's samples are checked, not trusted: an index outside the data gives no model
    bad = samples[:2].copy(); bad[0, 0] = 500; bad[1, 1] = -1
    assert np.isnan(gpu_ctx.two_view_minimal(model, sc["x1"], sc["x2"], sc["wh"], bad)).all()


@pytest.mark.parametrize("model", ["F", "H"])
def test_against_oracle_with_its_own_solver(gpu_ctx, oracle, model):
    """nothing the oracle is fed comes from the GPU: its own minimal solvers.  Tier 2 on every scene (the same solution of the scene),
    tier 1 where both runs end on the model of the same iteration (agreement to rounding)."""
    same_iter = 0
    cases = [(300, 81), (1000, 82), (600, 83), (1500, 84), (800, 85)]
    for k, (n, seed) in enumerate(cases):
        sc = tvh.scene(n, seed, planar=model == "H")
        got = gpu_ctx.two_view_acransac(model, sc["x1"], sc["x2"], sc["wh"], max_iteration=256, seed=90 + k)
        want = oracle.acransac(KIND[model], sc["x1"], sc["x2"], np.eye(3), _oracle_fit(oracle, model, sc), max_iteration=256, seed=90 + k, img_wh=sc["wh"])
        ref = oracle.acransac(KIND[model], sc["x1"], sc["x2"], np.eye(3), _device_fit(gpu_ctx, model, sc), max_iteration=256, seed=90 + k, img_wh=sc["wh"])
        assert want["found"] and got["M"] is not None
        a, b = set(got["inliers"].tolist()), set(want["inliers"].tolist())
        assert len(a & b) >= 0.95 * len(a | b), (len(a), len(b), len(a & b))
        assert 0.5 * want["error_max"] <= got["error_max"] <= 2.0 * want["error_max"]     # (another winning sample: another threshold, the same scale)
        if ref["best_iter"] == want["best_iter"] and got["iterations"] == want["iterations"]:
            same_iter += 1
            assert np.abs(tvh.unit(got["M"]) - tvh.unit(want["model"])).max() < 1e-6
            assert abs(got["error_max"] - want["error_max"]) <= 1e-6 * want["error_max"]
            assert abs(got["min_nfa"] - want["min_nfa"]) <= 1e-6 * abs(want["min_nfa"])
            assert len(a ^ b) <= max(1, n // 500)
    print("%s: scenes in which both runs picked the same iteration: %d of %d" % (model, same_iter, len(cases)))
    assert same_iter >= 2


@pytest.mark.parametrize("model", ["F", "H"])
def test_deterministic_and_seed_dependent(gpu_ctx, model):
    sc = tvh.scene(700, 91, planar=model == "H")
    a = gpu_ctx.two_view_acransac(model, sc["x1"], sc["x2"], sc["wh"], seed=5)
    b = gpu_ctx.two_view_acransac(model, sc["x1"], sc["x2"], sc["wh"], seed=5)
    c = gpu_ctx.two_view_acransac(model, sc["x1"], sc["x2"], sc["wh"], seed=6)
    assert np.array_equal(a["M"], b["M"]) and np.array_equal(a["inliers"], b["inliers"]) and a["min_nfa"] == b["min_nfa"]
    assert not np.array_equal(a["M"], c["M"])
    # the model 'E' of the same entry is clc_essential_acransac
    K = tvh.K_DEFAULT
    e1 = gpu_ctx.two_view_acransac("E", sc["x1"], sc["x2"], sc["wh"], K1=K, K2=K, seed=5)
    e2 = gpu_ctx.essential_acransac(sc["x1"], sc["x2"], K, K, sc["wh"], seed=5)
    assert np.array_equal(e1["M"], e2["E"]) and np.array_equal(e1["F"], e2["F"]) and np.array_equal(e1["inliers"], e2["inliers"])


@pytest.mark.parametrize("model", ["F", "H"])
@pytest.mark.parametrize("n_jobs", [3, 8])
def test_batches_equal_the_single_solves(gpu_ctx, model, n_jobs):
    """clc_two_view_acransac_batch: chains of their own (3 jobs) and lockstep rounds in shared launches (8 jobs) give every job the
    result of its single solve; a job without data does not disturb the others"""
    ctxs = [abi.Context(device=0, detector=False, matcher=False) for _ in range(n_jobs)]
    try:
        scs = [tvh.scene(300 + 170 * j, 100 + j, planar=model == "H") for j in range(n_jobs)]
        problems = [(sc["x1"], sc["x2"], None, None, sc["wh"], 200 + j) for j, sc in enumerate(scs)]
        problems[1] = (scs[1]["x1"][:3], scs[1]["x2"][:3], None, None, scs[1]["wh"], 201)
        res = abi.two_view_acransac_batch(ctxs, model, problems)
        for j, (r, p) in enumerate(zip(res, problems)):
            single = gpu_ctx.two_view_acransac(model, p[0], p[1], p[4], seed=p[5])
            assert r["status"] == 0
            if single["M"] is None:
                assert r["E"] is None and j == 1
                continue
            assert np.array_equal(r["E"], single["M"]) and np.array_equal(r["inliers"], single["inliers"])
            assert r["error_max"] == single["error_max"] and r["min_nfa"] == single["min_nfa"] and r["iterations"] == single["iterations"]
            assert np.array_equal(r["F"], single["F"])
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("model", ["F", "H"])
def test_against_the_golden_fixture(gpu_ctx, model):
    """tests/golden/twoview_models.npz (the oracle's filter with its own solvers, frozen): the device's models of the fixture's first
    sample are the fixture's to rounding, and the device's filter keeps the fixture's inliers (another null-space basis: the runs may
    end on different iterations, the solution of the scene is the same)"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "twoview_models.npz"))
    x1, x2, wh, seed = g[model + "_x1"], g[model + "_x2"], tuple(int(v) for v in g[model + "_wh"]), int(g[model + "_seed"])
    dev = [m for m in gpu_ctx.two_view_minimal(model, x1, x2, wh, g[model + "_first_sample"][None, :].astype(np.int32))[0] if not np.isnan(m).any()]
    gold = list(g[model + "_first_models"])
    assert len(dev) == len(gold)
    for a in dev:
        assert min(float(np.abs(tvh.unit(a) - tvh.unit(b)).max()) for b in gold) < 1e-6
    got = gpu_ctx.two_view_acransac(model, x1, x2, wh, max_iteration=128, seed=seed)
    a, b = set(got["inliers"].tolist()), set(g[model + "_inliers"].tolist())
    # (240 correspondences and a point-to-line residual: two runs that end on different samples differ by a handful of border points)
    assert got["M"] is not None and len(a & b) >= 0.9 * len(a | b)
    assert 0.5 * float(g[model + "_error_max"]) <= got["error_max"] <= 2.0 * float(g[model + "_error_max"])
