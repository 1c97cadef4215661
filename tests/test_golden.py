"""Oracle vs the committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).
feeder_ref.npz holds outputs of the REFERENCE's own KFAST.h / FeatureAngle.h run in the build
container; the others freeze the restatement."""
import os

import numpy as np

import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_k2nn_golden(oracle):
    g = np.load(os.path.join(G, "k2nn_512.npz"))
    m, b, s = oracle.k2nn(g["Q"], g["T"], 40, want_dist=True)
    assert np.array_equal(m, g["match40"]) and np.array_equal(b, g["best"]) and np.array_equal(s, g["second"])
    assert np.array_equal(oracle.k2nn(g["Q"], g["T"], 60), g["match60"])
    assert g["match40"][5] == -1            # planted exact duplicate pair (7, 400) -> tie -> rejected


def test_pyramid_and_clatch_golden(oracle):
    g = np.load(os.path.join(G, "clatch_160x120.npz"))
    pyr = oracle.pyramid(g["img"])
    for i in range(8):
        assert np.array_equal(pyr[i], g["level%d" % i])
    kps = g["kps"].reshape(-1).view(synth.KP_DTYPE)
    assert np.array_equal(oracle.clatch(pyr, kps), g["desc"])


def test_pnp_golden(oracle):
    g = np.load(os.path.join(G, "pnp_8x64.npz"))
    e = oracle.pnp_residuals(g["Rt"], g["X"], g["x"], g["K"])
    assert np.array_equal(e, g["err"])
    cnt, cost = oracle.pnp_score(e, 16.0)
    assert np.array_equal(cnt, g["count"]) and np.allclose(cost, g["cost"], rtol=1e-15)


def test_feeder_against_reference_outputs(oracle):
    g = np.load(os.path.join(G, "feeder_ref.npz"))
    for name in ("a", "b"):
        img = g["img_" + name]
        k = oracle.fast9(img, 40)
        assert np.array_equal(np.stack([k["x"], k["y"], k["score"].astype(np.int32)], 1), g["xys_" + name])
        ang = np.array([oracle.feature_angle(img, int(p["x"]), int(p["y"])) for p in k], dtype=np.float32)
        assert np.array_equal(ang.view(np.uint32), g["angle_" + name].view(np.uint32))


def test_twoview_models_golden(oracle):
    """the a-contrario filter under the seven-point / four-point models, the oracle's own solvers (tests/golden/make_golden_twoview.py):
    conditioning bit for bit, the first sample's models and the run's numbers to 1e-9 (libm's acos / cos / pow sit in the cubic),
    the discrete results exactly"""
    g = np.load(os.path.join(G, "twoview_models.npz"))
    for model, kind in (("F", 2), ("H", 3)):
        x1, x2, wh, seed = g[model + "_x1"], g[model + "_x2"], tuple(int(v) for v in g[model + "_wh"]), int(g[model + "_seed"])
        q1, q2 = oracle.tv_normalize(wh, x1), oracle.tv_normalize(wh, x2)
        assert np.array_equal(q1, g[model + "_q1"]) and np.array_equal(q2, g[model + "_q2"])
        smp = g[model + "_first_sample"]
        mods = oracle.seven_point(q1[smp], q2[smp]) if model == "F" else [oracle.four_point(q1[smp], q2[smp])]
        assert np.allclose(np.array(mods), g[model + "_first_models"], rtol=1e-9, atol=1e-12)
        fit = (lambda s: oracle.seven_point(q1[s], q2[s])) if model == "F" else (lambda s: [oracle.four_point(q1[s], q2[s])])
        r = oracle.acransac(kind, x1, x2, np.eye(3), fit, max_iteration=128, seed=seed, img_wh=wh)
        assert r["found"] and r["iterations"] == int(g[model + "_iterations"]) and r["best_iter"] == int(g[model + "_best_iter"])
        assert np.array_equal(r["inliers"], g[model + "_inliers"])
        assert np.allclose(r["model"], g[model + "_model"], rtol=1e-9, atol=1e-12)
        assert abs(r["error_max"] - float(g[model + "_error_max"])) <= 1e-9 * float(g[model + "_error_max"])
        assert abs(r["min_nfa"] - float(g[model + "_min_nfa"])) <= 1e-9 * abs(float(g[model + "_min_nfa"]))
        assert np.array_equal(np.array(r["samples"][0]), smp)
