"""CPU tests of the seven-point ('F') and four-point ('H') models (reference include/coloc/RobustMatcher.hpp:128-151, :188-239): the
oracle's own solvers (oracle/clc_oracle_twoview.c, one-sided Jacobi SVD) against numpy's SVD, the PRODUCT's statement
(coloc_amd/csrc/twoview_min.h, Householder null space; host build) against the oracle's, and the sequential a-contrario loop of the
oracle on the two kinds."""
import numpy as np
import pytest

import twoview_host as tvh


def _null_np(A, k):
    return np.linalg.svd(A)[2][-k:]


def _seven_rows(q1, q2):
    return np.array([[b[0] * a[0], b[0] * a[1], b[0], b[1] * a[0], b[1] * a[1], b[1], a[0], a[1], 1.0] for a, b in zip(q1, q2)])


def test_normaliser_is_the_image_size_conditioning(oracle):
    wh = (1280, 720)
    d = 1.0 / np.sqrt(1280.0 * 720.0)
    x = np.array([[0.0, 0.0], [1280.0, 720.0], [640.0, 360.0], [17.5, 701.25]])
    want = x * d + np.array([-0.5 * 1280 * d, -0.5 * 720 * d])
    assert np.array_equal(oracle.tv_normalize(wh, x), want)
    assert np.array_equal(tvh.normalizer(wh), np.array([d, -0.5 * 1280 * d, -0.5 * 720 * d]))
    # un-normalising: F = T^T Fn T, H = T^-1 Hn T, the same bits from the two statements
    T = np.array([[d, 0, -0.5 * 1280 * d], [0, d, -0.5 * 720 * d], [0, 0, 1]])
    rng = np.random.default_rng(3)
    M = rng.standard_normal((3, 3))
    assert np.allclose(oracle.tv_unnormalize(False, wh, M), T.T @ M @ T, rtol=1e-13, atol=1e-15)
    assert np.allclose(oracle.tv_unnormalize(True, wh, M), np.linalg.inv(T) @ M @ T, rtol=1e-12, atol=1e-9)
    assert np.array_equal(oracle.tv_unnormalize(False, wh, M), tvh.unnormalize(False, wh, M))
    assert np.array_equal(oracle.tv_unnormalize(True, wh, M), tvh.unnormalize(True, wh, M))


def test_seven_point_statements_agree_and_solve_the_system(oracle):
    checked = 0
    for seed in range(300):
        sc = tvh.scene(7, seed, outlier_frac=0.0, noise=0.0)
        q1, q2 = oracle.tv_normalize(sc["wh"], sc["x1"]), oracle.tv_normalize(sc["wh"], sc["x2"])
        Fo = oracle.seven_point(q1, q2)
        Fp = tvh.seven_point(q1, q2)
        assert len(Fo) in (1, 3) and len(Fo) == len(Fp)
        A = _seven_rows(q1, q2)
        N = _null_np(A, 2)
        for F in Fo + Fp:
            Fu = tvh.unit(F)
            assert np.abs(A @ Fu).max() < 1e-10                                # the seven epipolar equations
            assert abs(np.linalg.det(Fu.reshape(3, 3))) < 1e-10                # rank 2
            assert np.linalg.norm(Fu - N.T @ (N @ Fu)) < 1e-10                 # inside numpy's null space
        # root by root the same matrices (ascending roots of the same pencil up to its parametrisation: compare as sets)
        for F in Fo:
            assert min(np.abs(tvh.unit(F) - tvh.unit(G)).max() for G in Fp) < 1e-7
        # one of them is the scene's
        Ft = tvh.unit(_normalised_f(sc))
        assert min(np.abs(tvh.unit(F) - Ft).max() for F in Fp) < 1e-6
        checked += len(Fo)
    assert checked >= 400


def _normalised_f(sc):
    """the scene's F in normalised coordinates: Fn = T^-T F T^-1"""
    w, h = sc["wh"]
    d = 1.0 / np.sqrt(float(w) * h)
    Ti = np.linalg.inv(np.array([[d, 0, -0.5 * w * d], [0, d, -0.5 * h * d], [0, 0, 1]]))
    return (Ti.T @ sc["F"] @ Ti).reshape(9)


def test_four_point_statements_agree_and_map_the_points(oracle):
    for seed in range(300):
        sc = tvh.scene(4, seed, planar=True, outlier_frac=0.0, noise=0.0)
        q1, q2 = oracle.tv_normalize(sc["wh"], sc["x1"]), oracle.tv_normalize(sc["wh"], sc["x2"])
        Ho, Hp = oracle.four_point(q1, q2), tvh.four_point(q1, q2)
        assert np.abs(tvh.unit(Ho) - tvh.unit(Hp)).max() < 1e-8
        for Hn in (Ho, Hp):
            assert oracle.tv_residuals(3, Hn, q1, q2).max() < 1e-18
            Hpix = oracle.tv_unnormalize(True, sc["wh"], Hn)
            assert np.abs(tvh.unit(Hpix) - tvh.unit(sc["H"])).max() < 1e-7


def test_cubic_roots_closed_form():
    rng = np.random.default_rng(5)
    for _ in range(2000):
        a, b, c = rng.uniform(-3, 3, 3)
        x = tvh.cubic(a, b, c)
        want = np.roots([1.0, a, b, c])
        want = np.sort(want[np.abs(want.imag) < 1e-9].real)
        if len(want) != len(x):                       # a pair of roots at the edge of being real
            continue
        assert np.allclose(x, want, rtol=1e-7, atol=1e-7)
        assert np.all(np.diff(x) >= 0)
    assert np.allclose(tvh.cubic(-6.0, 11.0, -6.0), [1.0, 2.0, 3.0], atol=1e-12)
    assert np.allclose(tvh.cubic(0.0, 0.0, -8.0), [2.0], atol=1e-12)


def test_residuals_are_the_textbook_distances(oracle):
    sc = tvh.scene(50, 9, planar=True, outlier_frac=0.0, noise=0.5)
    q1, q2 = oracle.tv_normalize(sc["wh"], sc["x1"]), oracle.tv_normalize(sc["wh"], sc["x2"])
    Fn = _normalised_f(sc)
    e = oracle.tv_residuals(2, Fn, q1, q2)
    F = Fn.reshape(3, 3)
    l = np.c_[q1, np.ones(50)] @ F.T                      # epipolar lines in image 2
    want = (np.sum(l * np.c_[q2, np.ones(50)], axis=1) ** 2) / (l[:, 0] ** 2 + l[:, 1] ** 2)
    assert np.allclose(e, want, rtol=1e-12, atol=0)
    w, h = sc["wh"]
    d = 1.0 / np.sqrt(float(w) * h)
    T = np.array([[d, 0, -0.5 * w * d], [0, d, -0.5 * h * d], [0, 0, 1]])
    Hn = T @ sc["H"] @ np.linalg.inv(T)
    e = oracle.tv_residuals(3, Hn, q1, q2)
    y = np.c_[q1, np.ones(50)] @ Hn.T
    want = np.sum((q2 - y[:, :2] / y[:, 2:]) ** 2, axis=1)
    assert np.allclose(e, want, rtol=1e-10, atol=1e-18)
    # in pixels: the point-to-point residual scales with d^2
    y = np.c_[sc["x1"], np.ones(50)] @ sc["H"].T
    pix = np.sum((sc["x2"] - y[:, :2] / y[:, 2:]) ** 2, axis=1)
    assert np.allclose(e, pix * d * d, rtol=1e-8)


@pytest.mark.parametrize("kind,planar", [(2, False), (3, True)])
def test_sequential_loop_finds_the_scene(oracle, kind, planar):
    """the oracle's a-contrario loop with its own minimal solvers on a scene with 30 % outliers: the inliers it keeps are the scene's,
    the model brought back to pixels is the scene's, the threshold is a few noise sigmas (pixels)"""
    for seed in (11, 12, 13):
        n = 400
        sc = tvh.scene(n, seed, planar=planar)
        q1, q2 = oracle.tv_normalize(sc["wh"], sc["x1"]), oracle.tv_normalize(sc["wh"], sc["x2"])

        def fit(sample):
            if kind == 2:
                return oracle.seven_point(q1[sample], q2[sample])
            return [oracle.four_point(q1[sample], q2[sample])]

        res = oracle.acransac(kind, sc["x1"], sc["x2"], np.eye(3), fit, max_iteration=256, seed=seed, img_wh=sc["wh"])
        assert res["found"] and res["min_nfa"] < -50
        true_in = np.ones(n, bool); true_in[sc["outliers"]] = False
        got = np.zeros(n, bool); got[res["inliers"]] = True
        assert (got & true_in).sum() >= 0.9 * true_in.sum()
        assert (got & ~true_in).sum() <= (0.2 if kind == 2 else 0.05) * len(sc["outliers"]) + 2      # (a line catches more strays than a point)
        if kind == 2:
            assert np.abs(tvh.unit(res["model"]) - tvh.unit(sc["F"])).max() < 0.05
        else:
            # a minimal-sample homography from noisy points: judged by where it sends the scene's points (pixels)
            y = np.c_[sc["x1"], np.ones(n)] @ res["model"].reshape(3, 3).T
            assert np.median(np.linalg.norm(sc["x2"] - y[:, :2] / y[:, 2:], axis=1)[true_in]) < 1.5
        assert 0.3 < res["error_max"] < 6.0                                     # pixels: sqrt(e) / N2(0,0)
        assert all(len(s) == (7 if kind == 2 else 4) for s in res["samples"])


def test_sequential_loop_no_model_cases(oracle):
    rng = np.random.default_rng(2)
    x1 = np.c_[rng.uniform(0, 1280, 60), rng.uniform(0, 720, 60)]
    x2 = np.c_[rng.uniform(0, 1280, 60), rng.uniform(0, 720, 60)]
    q1, q2 = oracle.tv_normalize((1280, 720), x1), oracle.tv_normalize((1280, 720), x2)
    res = oracle.acransac(3, x1, x2, np.eye(3), lambda s: [oracle.four_point(q1[s], q2[s])], max_iteration=64, seed=1, img_wh=(1280, 720))
    assert not res["found"] and len(res["inliers"]) == 0
    # not more data than the sample: nothing to do (ACRANSAC returns at once)
    res = oracle.acransac(2, x1[:7], x2[:7], np.eye(3), lambda s: [], max_iteration=64, seed=1, img_wh=(1280, 720))
    assert not res["found"] and res["iterations"] == 0
