"""GPU a-contrario RANSAC (clc_pnp_acransac / clc_essential_acransac, coloc_amd/csrc/acransac.hip) against the sequential
oracle (oracle/clc_oracle_acr.c): the batched rounds must reproduce the iteration-by-iteration loop of OpenMVG's
ACRANSAC as the reference calls it (Localizer.hpp:82-93: error_max = +inf, 256 iterations; RobustMatcher.hpp:161-171) --
same model, same inlier list in the same order, same threshold, same number of iterations, and the same NFA to NFA_RTOL:
since round 6 the oracle shares no code with the product (its own statement of the sampler, libm's log10 where the GPU
evaluates the portable log10 of clc_acr.h, which is within 2 ulp of libm's -- tests/test_acransac.py), so the discrete
results are compared exactly and the one value that is a sum of logarithms to a stated tolerance.  The oracle's minimal
solver is a callback: into the product's own P3P / five-point kernels (checked against independent solvers in
test_gpu_pnp.py / test_gpu_epipolar.py) where exactly the a-contrario machinery is compared, into HOST builds of the
solvers where nothing the oracle is fed may come from the GPU (test_pose_against_host_solved_oracle,
test_essential_against_host_solved_oracle)."""
import math

import numpy as np
import pytest

import synth
from test_gpu_epipolar import _two_view

pytestmark = pytest.mark.gpu
# log10 NFA = loge0 + (logalpha0 + mult log10(e_k + eps)) (k - m) + tables: a 2-ulp difference in the one logarithm that depends on the
# residual is scaled by (k - m) like the term itself -- a few 1e-16 relative; 1e-12 leaves three decades of room
NFA_RTOL = 1e-12


def _same_nfa(a, b):
    return (math.isinf(a) and math.isinf(b)) or abs(a - b) <= NFA_RTOL * abs(b)


def _p3p_fit(ctx, sc):
    def fit(sample):
        h = ctx.pnp_p3p(sc["X"], sc["x"], sc["K"], np.array([sample], dtype=np.int32))[0]
        return [m for m in h if not np.isnan(m).any()]
    return fit


def _check_pose(ctx, oracle, sc, max_it, seed, precision=float("inf")):
    got = ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], max_iteration=max_it, seed=seed, precision=precision)
    want = oracle.acransac(0, sc["X"], sc["x"], sc["K"], _p3p_fit(ctx, sc), max_iteration=max_it, seed=seed, precision=precision)
    assert (got["Rt"] is not None) == want["found"]
    assert got["iterations"] == want["iterations"]
    assert _same_nfa(got["min_nfa"], want["min_nfa"])
    assert np.array_equal(got["inliers"], want["inliers"].astype(np.int32))
    if want["found"]:
        assert np.array_equal(got["Rt"].reshape(-1), want["model"])
        assert got["error_max"] == want["error_max"]
        m = np.zeros(len(sc["X"]), bool); m[want["inliers"]] = True
        assert np.array_equal(got["mask"], m)
    else:
        assert not got["mask"].any()
    return got, want


@pytest.mark.parametrize("n", [200, 1000, 5000])
def test_pose_equals_sequential_oracle_config2_sizes(gpu_ctx, oracle, n):
    """BASELINE config[2] sizes, the reference's settings (256 iterations, precision +inf)."""
    sc = synth.pnp_scene(n, seed=4000 + n)
    got, want = _check_pose(gpu_ctx, oracle, sc, 256, seed=1)
    assert want["found"] and want["min_nfa"] < 0
    # the a-contrario threshold separates the planted inliers without being told a threshold
    assert (got["mask"] & sc["inliers"]).sum() >= 0.9 * sc["inliers"].sum()      # the a-contrario cut drops the noise tail
    assert (got["mask"] & ~sc["inliers"]).sum() <= 0.02 * n
    true = np.concatenate([sc["R"], sc["t"][:, None]], 1)
    assert np.abs(got["Rt"] - true).max() < 0.05
    assert 0.5 < got["error_max"] < 4.0                      # pixels; the noise is 0.5 px


@pytest.mark.parametrize("n,outl,max_it,seed", [(300, 0.6, 256, 2), (800, 0.75, 256, 3), (64, 0.2, 40, 4), (10, 0.0, 30, 5),
                                                (4, 0.0, 20, 6), (2000, 0.5, 100, 7), (513, 0.4, 256, 8)])
def test_pose_other_shapes_and_outlier_rates(gpu_ctx, oracle, n, outl, max_it, seed):
    """Low inlier rates make the first phase long (several rounds, larger batches); tiny n exercises n = m + 1."""
    sc = synth.pnp_scene(n, seed=4100 + seed, outlier_frac=outl)
    _check_pose(gpu_ctx, oracle, sc, max_it, seed)


def test_pose_many_seeds_same_bits_as_p3p_kernel(gpu_ctx, oracle):
    """acr_round_kernel solves its samples in place (p3p_sample_root inlined into it); the oracle gets its poses from p3p_kernel (the
    same body inlined into another kernel).  Forty runs with different seeds, sizes and outlier rates: the winning pose -- a direct
    output of that body -- and everything selected on the residuals of every evaluated pose must agree bit for bit."""
    for k in range(40):
        n = (150, 260, 700, 1500)[k % 4]
        sc = synth.pnp_scene(n, seed=7000 + k, outlier_frac=(0.2, 0.45, 0.7)[k % 3])
        _check_pose(gpu_ctx, oracle, sc, 256 if k % 2 else 90, seed=100 + k)


def test_pose_against_host_solved_oracle(gpu_ctx, oracle):
    """Parity that does not lean on GPU-made inputs: the sequential oracle is handed minimal solutions from the HOST build of
    coloc_amd/csrc/p3p.h (tests/host/p3p_host_lib.cpp: IEEE division and the host compiler's contraction instead of v_rcp_f64 + Newton
    and the device compiler's).  Host and device poses agree to rounding, not bit for bit (checked: <= 1e-6 on every pose slot,
    ill-conditioned samples included), and an a-contrario run is a discrete process -- one correspondence moving across a threshold
    changes the index set the next samples are drawn from -- so the statement has two tiers, both asserted:
      * scenes where the two runs pick the model of the SAME iteration: pose within 1e-7, threshold and NFA within 1e-6 relative,
        inlier sets equal up to 0.2 % of the correspondences;
      * every scene: both results are the same solution of the scene -- inlier sets overlap >= 97 % (Jaccard), both poses within 0.05
        of the pose the scene was generated from, thresholds within 25 %."""
    import p3p_host
    same_iter = 0
    for k, (n, outl) in enumerate([(200, 0.3), (1000, 0.3), (5000, 0.3), (300, 0.6), (800, 0.7), (1500, 0.45), (64, 0.2), (2000, 0.5)]):
        sc = synth.pnp_scene(n, seed=8200 + k, outlier_frac=outl)
        seen = {"max_dev": 0.0, "n": 0, "slot_mismatch": 0}

        def fit(sample):           # (runs inside a ctypes callback: nothing here may raise)
            h = p3p_host.sample_poses(sc["X"], sc["x"], sc["K"], sample)
            d = gpu_ctx.pnp_p3p(sc["X"], sc["x"], sc["K"], np.array([sample], dtype=np.int32))[0].reshape(4, 3, 4)
            ok_h, ok_d = ~np.isnan(h).any(axis=(1, 2)), ~np.isnan(d).any(axis=(1, 2))
            both = ok_h & ok_d
            seen["slot_mismatch"] += int((ok_h != ok_d).sum())                 # a root on the edge of validity may exist on one side only
            if both.any():
                seen["max_dev"] = max(seen["max_dev"], float(np.abs(h[both] - d[both]).max()))
                seen["n"] += int(both.sum())
            return [m.reshape(12) for m in h[ok_h]]

        got = gpu_ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], max_iteration=256, seed=11 + k)
        want = oracle.acransac(0, sc["X"], sc["x"], sc["K"], fit, max_iteration=256, seed=11 + k)
        assert seen["n"] >= 20 and seen["max_dev"] < 1e-6 and seen["slot_mismatch"] <= 2, seen    # host and device solver: rounding-level agreement
        assert want["found"] and got["Rt"] is not None
        a, b = set(got["inliers"].tolist()), set(want["inliers"].tolist())
        true = np.concatenate([sc["R"], sc["t"][:, None]], 1).reshape(-1)
        dev = float(np.abs(got["Rt"].reshape(-1) - want["model"]).max())
        # tier 2: the same solution of the scene
        assert len(a & b) >= 0.97 * len(a | b), (len(a), len(b), len(a & b))
        assert np.abs(got["Rt"].reshape(-1) - true).max() < 0.05 and np.abs(want["model"] - true).max() < 0.05
        assert abs(got["error_max"] - want["error_max"]) <= 0.25 * want["error_max"]
        # tier 1: the same winning iteration -> agreement to rounding
        gpu_ref = oracle.acransac(0, sc["X"], sc["x"], sc["K"], _p3p_fit(gpu_ctx, sc), max_iteration=256, seed=11 + k)      # (for its best_iter)
        if gpu_ref["best_iter"] == want["best_iter"] and got["iterations"] == want["iterations"]:
            same_iter += 1
            assert dev < 1e-7
            assert abs(got["error_max"] - want["error_max"]) <= 1e-6 * want["error_max"]
            assert abs(got["min_nfa"] - want["min_nfa"]) <= 1e-6 * abs(want["min_nfa"])
            assert len(a ^ b) <= max(1, n // 500), (len(a), len(b), len(a ^ b))
        print("n %5d: %4d pose slots, host/device solver differ by <= %.1e; winning iteration %d / %d, poses differ by %.1e, inlier sets by %d of %d"
              % (n, seen["n"], seen["max_dev"], gpu_ref["best_iter"], want["best_iter"], dev, len(a ^ b), len(a | b)))
    print("scenes in which both runs picked the same iteration: %d of 8" % same_iter)
    assert same_iter >= 4


def test_pose_duplicate_and_nearly_equal_residuals(gpu_ctx, oracle):
    """The GPU sorts 64-bit words made of the top 51 residual bits + the index and repairs the order exactly afterwards.
    Duplicated correspondences (equal residuals: index decides) and copies whose pixel differs by one ulp (residuals
    that agree in their top bits but not exactly: the repair path) must still give the oracle's order."""
    sc = synth.pnp_scene(400, seed=4242, outlier_frac=0.3)
    X, x = sc["X"].copy(), sc["x"].copy()
    inl = np.flatnonzero(sc["inliers"])
    for k in range(60):                                        # exact duplicates, scattered over the index range
        X[399 - k] = X[inl[k]]; x[399 - k] = x[inl[k]]
    for k in range(60, 120):                                   # one-ulp copies in both directions
        X[399 - k] = X[inl[k]]
        x[399 - k] = np.nextafter(x[inl[k]], np.inf if k % 2 else -np.inf)
    sc2 = dict(sc, X=X, x=x)
    got, want = _check_pose(gpu_ctx, oracle, sc2, 64, seed=12)
    assert want["found"]


def test_pose_no_model_cases(gpu_ctx, oracle):
    sc = synth.pnp_scene(3, seed=1, outlier_frac=0.0)
    got = gpu_ctx.pnp_acransac(sc["X"], sc["x"], sc["K"])
    assert got["Rt"] is None and got["iterations"] == 0 and len(got["inliers"]) == 0
    rng = np.random.default_rng(3)                           # pure noise: every NFA >= 0, the reserve is spent looking
    sc = dict(X=rng.uniform(-5, 5, (150, 3)) + [0, 0, 12], x=np.stack([rng.uniform(0, 1280, 150), rng.uniform(0, 720, 150)], 1),
              K=np.array([[1000.0, 0, 640], [0, 1000.0, 360], [0, 0, 1]]))
    got, want = _check_pose(gpu_ctx, oracle, sc, 60, seed=9)
    assert got["Rt"] is None and want["min_nfa"] >= 0 and got["iterations"] == 60


def test_pose_upper_bound_mode(gpu_ctx, oracle):
    """A finite precision (pixels^2) is OpenMVG's upper bound on the inlier residual: the NFA scan stops there."""
    sc = synth.pnp_scene(600, seed=4555, outlier_frac=0.4)
    got, want = _check_pose(gpu_ctx, oracle, sc, 128, seed=11, precision=9.0)
    assert want["found"] and got["error_max"] <= 3.0 + 1e-9
    got2, _ = _check_pose(gpu_ctx, oracle, sc, 128, seed=11, precision=0.25)
    assert got2["error_max"] <= 0.5 + 1e-9 and len(got2["inliers"]) < len(got["inliers"])


def test_pose_is_deterministic_and_seed_dependent(gpu_ctx):
    sc = synth.pnp_scene(1000, seed=4777)
    a = gpu_ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], seed=5)
    b = gpu_ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], seed=5)
    c = gpu_ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], seed=6)
    assert np.array_equal(a["Rt"], b["Rt"]) and np.array_equal(a["inliers"], b["inliers"]) and a["min_nfa"] == b["min_nfa"]
    assert c["Rt"] is not None and (c["mask"] & sc["inliers"]).sum() >= 0.9 * sc["inliers"].sum()


def test_localize_ac_is_acransac_then_refine(gpu_ctx):
    sc = synth.pnp_scene(1000, seed=4999)
    a = gpu_ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], seed=3)
    r = gpu_ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], seed=3, refine=True)
    assert np.array_equal(a["inliers"], r["inliers"]) and a["error_max"] == r["error_max"]
    Rt2, cov2, rmse2, _ = gpu_ctx.pnp_refine(sc["X"], sc["x"], sc["K"], a["Rt"], mask=a["mask"])
    assert np.allclose(r["Rt"], Rt2, atol=1e-9) and np.allclose(r["cov"], cov2, rtol=1e-6, atol=1e-12) and abs(r["rmse"] - rmse2) < 1e-9
    true = np.concatenate([sc["R"], sc["t"][:, None]], 1)
    assert np.abs(r["Rt"] - true).max() < np.abs(a["Rt"] - true).max() + 1e-6 and np.abs(r["Rt"] - true).max() < 5e-3


def _f_from_e(E, K1, K2):
    """F = K2^-T E K1^-1 in the operation order of fivept_kernel (coloc_amd/csrc/pnp.hip), plain IEEE doubles."""
    def kinv(K):
        fx, sk, cx, fy, cy = float(K[0, 0]), float(K[0, 1]), float(K[0, 2]), float(K[1, 1]), float(K[1, 2])
        return [1.0 / fx, -sk / (fx * fy), (sk * cy - cx * fy) / (fx * fy), 0.0, 1.0 / fy, -cy / fy, 0.0, 0.0, 1.0]
    A1, A2 = kinv(K1), kinv(K2)
    e = [float(v) for v in E.reshape(9)]
    T = [0.0] * 9
    for r in range(3):
        for c in range(3):
            T[3 * r + c] = e[3 * r] * A1[c] + e[3 * r + 1] * A1[3 + c] + e[3 * r + 2] * A1[6 + c]
    F = [0.0] * 9
    for r in range(3):
        for c in range(3):
            F[3 * r + c] = A2[r] * T[c] + A2[3 + r] * T[3 + c] + A2[6 + r] * T[6 + c]
    return np.array(F)


@pytest.mark.parametrize("n,seed", [(300, 21), (1000, 22)])
def test_essential_equals_sequential_oracle(gpu_ctx, oracle, n, seed):
    x1, x2, Ftrue, out = _two_view(n, seed=seed)
    K = synth.pnp_scene(5, seed=seed)["K"]

    def fit(sample):
        Es = gpu_ctx.essential_fivepoint(x1, x2, K, K, np.array([sample], dtype=np.int32))[0]
        return [np.concatenate([_f_from_e(E, K, K), E]) for E in Es if not np.isnan(E).any()]

    got = gpu_ctx.essential_acransac(x1, x2, K, K, (1280, 720), max_iteration=256, seed=seed)
    want = oracle.acransac(1, x1, x2, K, fit, max_iteration=256, seed=seed, img_wh=(1280, 720))
    assert want["found"] and got["E"] is not None
    assert got["iterations"] == want["iterations"] and _same_nfa(got["min_nfa"], want["min_nfa"]) and got["error_max"] == want["error_max"]
    assert np.array_equal(got["inliers"], want["inliers"].astype(np.int32))
    assert np.array_equal(got["F"].reshape(-1), want["model"][:9]) and np.array_equal(got["E"].reshape(-1), want["model"][9:])
    true_in = np.ones(n, bool); true_in[out] = False
    assert (got["mask"] & true_in).sum() >= 0.9 * true_in.sum() and (got["mask"] & ~true_in).sum() <= 0.1 * len(out) + 2
    Fn = got["F"] / np.linalg.norm(got["F"])
    assert min(np.abs(Fn - Ftrue).max(), np.abs(Fn + Ftrue).max()) < 0.05


def test_essential_against_host_solved_oracle(gpu_ctx, oracle):
    """The five-point counterpart of test_pose_against_host_solved_oracle: the sequential oracle is handed essential
    matrices from the HOST build of the sequential five-point statement (coloc_amd/csrc/fivept.h through tests/host/fivept_host_lib.cpp,
    g++), not from fivept_kernel (the wave-cooperative form, coloc_amd/csrc/fivept_wave.h).  The two solvers find the same solutions to
    rounding -- checked per sample: every host solution has a device solution within 1e-6 (unit Frobenius norm, sign fixed), a root at
    the edge of validity may exist on one side only -- but in an order of their own, and an a-contrario run is a discrete process, so,
    as for the resection, two tiers:
      * every scene: both runs found the same solution of the scene -- inlier sets overlap >= 97 % (Jaccard), both F within 0.05 of the
        scene's (unit norm, sign fixed), thresholds within 25 %;
      * scenes where both runs pick the model of the SAME iteration: F within 1e-6, threshold and NFA within 1e-6 relative, inlier
        sets equal up to 0.2 % of the correspondences."""
    import fivept_host

    def unit(E):
        E = np.asarray(E, dtype=np.float64).reshape(9)
        E = E / np.linalg.norm(E)
        return E if E[np.argmax(np.abs(E))] > 0 else -E

    same_iter = 0
    for k, (n, seed) in enumerate([(300, 21), (1000, 22), (600, 23), (1500, 24), (800, 25)]):
        x1, x2, Ftrue, out = _two_view(n, seed=seed)
        K = synth.pnp_scene(5, seed=seed)["K"]
        Kinv = np.linalg.inv(K)
        q1 = (np.c_[x1, np.ones(n)] @ Kinv.T)[:, :2]
        q2 = (np.c_[x2, np.ones(n)] @ Kinv.T)[:, :2]
        seen = {"host": 0, "matched": 0, "max_dev": 0.0}

        def fit_host(sample):           # (runs inside a ctypes callback: nothing here may raise)
            Es = fivept_host.solve(q1[sample], q2[sample])
            dev = [unit(E) for E in gpu_ctx.essential_fivepoint(x1, x2, K, K, np.array([sample], dtype=np.int32))[0] if not np.isnan(E).any()]
            for E in Es:
                seen["host"] += 1
                d = min((float(np.abs(unit(E) - D).max()) for D in dev), default=1.0)
                if d < 1e-6:
                    seen["matched"] += 1
                    seen["max_dev"] = max(seen["max_dev"], d)
            return [np.concatenate([_f_from_e(E, K, K), E.reshape(9)]) for E in Es]

        def fit_gpu(sample):
            Es = gpu_ctx.essential_fivepoint(x1, x2, K, K, np.array([sample], dtype=np.int32))[0]
            return [np.concatenate([_f_from_e(E, K, K), E]) for E in Es if not np.isnan(E).any()]

        got = gpu_ctx.essential_acransac(x1, x2, K, K, (1280, 720), max_iteration=256, seed=30 + k)
        want = oracle.acransac(1, x1, x2, K, fit_host, max_iteration=256, seed=30 + k, img_wh=(1280, 720))
        assert seen["host"] >= 40 and seen["matched"] >= 0.97 * seen["host"], seen          # the two solvers: the same solutions, to rounding
        assert want["found"] and got["E"] is not None
        a, b = set(got["inliers"].tolist()), set(want["inliers"].tolist())
        # tier 2: the same solution of the scene
        assert len(a & b) >= 0.97 * len(a | b), (len(a), len(b), len(a & b))
        Ft = unit(Ftrue)
        assert np.abs(unit(got["F"]) - Ft).max() < 0.05 and np.abs(unit(want["model"][:9]) - Ft).max() < 0.05
        assert abs(got["error_max"] - want["error_max"]) <= 0.25 * want["error_max"]
        # tier 1: the same winning iteration -> agreement to rounding
        gpu_ref = oracle.acransac(1, x1, x2, K, fit_gpu, max_iteration=256, seed=30 + k, img_wh=(1280, 720))        # (for its best_iter)
        dev = float(np.abs(unit(got["F"]) - unit(want["model"][:9])).max())
        if gpu_ref["best_iter"] == want["best_iter"] and got["iterations"] == want["iterations"]:
            same_iter += 1
            assert dev < 1e-6
            assert abs(got["error_max"] - want["error_max"]) <= 1e-6 * want["error_max"]
            assert abs(got["min_nfa"] - want["min_nfa"]) <= 1e-6 * abs(want["min_nfa"])
            assert len(a ^ b) <= max(1, n // 500), (len(a), len(b), len(a ^ b))
        print("n %5d: %4d host solutions, %d with a device solution within 1e-6 (max %.1e); winning iteration %d / %d, F differs by %.1e, inlier sets by %d of %d"
              % (n, seen["host"], seen["matched"], seen["max_dev"], gpu_ref["best_iter"], want["best_iter"], dev, len(a ^ b), len(a | b)))
    print("scenes in which both runs picked the same iteration: %d of 5" % same_iter)
    assert same_iter >= 2


@pytest.mark.parametrize("n,max_it,seed", [(3000, 64, 21), (4096, 64, 22), (8192, 48, 23), (8193, 48, 24), (12000, 48, 25), (16384, 32, 26)])
def test_pose_large_correspondence_sets(gpu_ctx, oracle, n, max_it, seed):
    """Every per-thread element count of the per-model sort: 4 (2049..4096), 8 (..8192) and -- new in round 3 -- 16 elements per
    thread (8193..16384: a 10k-keypoint pair yields ~9.3k accepted matches, more than round 2's cap of 8192).  Same result as the
    sequential oracle, duplicates included so that the exact re-sort runs too."""
    sc = synth.pnp_scene(n, seed=4300 + seed, outlier_frac=0.4)
    X, x = sc["X"].copy(), sc["x"].copy()
    inl = np.flatnonzero(sc["inliers"])
    for k in range(40):                                        # duplicates + one-ulp copies, scattered: the exact re-sort path
        X[n - 1 - 7 * k] = X[inl[k]]
        x[n - 1 - 7 * k] = x[inl[k]] if k % 2 else np.nextafter(x[inl[k]], np.inf)
    _check_pose(gpu_ctx, oracle, dict(sc, X=X, x=x), max_it, seed)


def test_pose_capacity_is_reported(gpu_ctx):
    from coloc_amd.abi import CLCError
    sc = synth.pnp_scene(16385, seed=1)
    with pytest.raises(CLCError):
        gpu_ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], max_iteration=8, seed=1)


def test_essential_on_more_than_8192_matches(gpu_ctx, oracle):
    """The two-view filter on a 10k-keypoint pair's worth of matches (the size bench.py's pair produces)."""
    x1, x2 = _two_view(9300, seed=77)[:2]
    K = synth.pnp_scene(5, seed=77)["K"]
    got = gpu_ctx.essential_acransac(x1, x2, K, K, (1280, 720), max_iteration=32, seed=5)
    assert got["E"] is not None and len(got["inliers"]) > 0.5 * 9300


def test_batched_localisations_equal_the_single_solves(gpu_ctx, oracle):
    """clc_pnp_localize_ac_batch (BASELINE config[2]: one pose per camera, all at once): eight independent a-contrario solves of different
    sizes, outlier rates and seeds, each on a context of its own and all driven by one host thread, must each give exactly what the
    single-solve entry gives -- model, covariance, inlier list in order, threshold -- with and without refinement; an empty and a tiny
    problem ride along; sharing a context between two jobs is refused."""
    from coloc_amd import Context
    from coloc_amd.abi import pnp_localize_batch, CLCError
    shapes = [(200, 0.3), (1000, 0.3), (5000, 0.4), (300, 0.6), (64, 0.2), (1500, 0.5), (3, 0.0), (800, 0.7)]
    scenes = [synth.pnp_scene(max(n, 4), seed=8600 + k, outlier_frac=o) for k, (n, o) in enumerate(shapes)]
    probs = [(sc["X"][:n], sc["x"][:n], sc["K"]) for sc, (n, _) in zip(scenes, shapes)]
    ctxs = [Context(device=0, detector=False, matcher=False) for _ in shapes]
    seeds = [31 + k for k in range(len(shapes))]
    try:
        for refine in (False, True):
            got = pnp_localize_batch(ctxs, probs, max_iteration=256, seeds=seeds, refine=refine)
            for k, (X, x, K) in enumerate(probs):
                want = gpu_ctx.pnp_acransac(X, x, K, max_iteration=256, seed=seeds[k], refine=refine)
                g = got[k]
                assert (g["Rt"] is None) == (want["Rt"] is None), k
                assert np.array_equal(g["inliers"], want["inliers"]) and np.array_equal(g["mask"], want["mask"]), k
                assert g["error_max"] == want["error_max"], k
                if want["Rt"] is not None:
                    assert np.array_equal(g["Rt"], want["Rt"]), k
                    if refine:
                        assert np.array_equal(g["cov"], want["cov"]) and g["rmse"] == want["rmse"], k
                    else:
                        assert g["iterations"] == want["iterations"], k
        assert got[6]["Rt"] is None and len(got[6]["inliers"]) == 0           # n = 3: no model (nData <= sizeSample)
        # the same again on the same contexts (their state is reused from batch to batch)
        again = pnp_localize_batch(ctxs, probs, max_iteration=256, seeds=seeds, refine=True)
        assert all(np.array_equal(a["inliers"], b["inliers"]) for a, b in zip(again, got))
        with pytest.raises(CLCError):
            pnp_localize_batch([ctxs[0], ctxs[0]], probs[:2], seeds=seeds[:2])
    finally:
        for c in ctxs:
            c.close()


def test_batched_localisations_one_bad_job_does_not_take_the_others_down(gpu_ctx):
    """clc_pnp_localize_ac_batch with a job the entry refuses (more than 16 384 correspondences) between two good ones: the call reports
    the failure (first failing status), the good jobs are solved all the same and equal the single solves, and the contexts stay usable."""
    import ctypes as C
    from coloc_amd import Context
    from coloc_amd.abi import PoseJob, load_library, CLC_OK
    lib = load_library()
    scs = [synth.pnp_scene(600, seed=9100), synth.pnp_scene(17000, seed=9101), synth.pnp_scene(900, seed=9102)]
    ctxs = [Context(device=0, detector=False, matcher=False) for _ in scs]
    try:
        jobs, keep = (PoseJob * 3)(), []
        for k, sc in enumerate(scs):
            X, x, K = (np.ascontiguousarray(sc[n], dtype=np.float64) for n in ("X", "x", "K"))
            N = X.shape[0]
            Rt, cov, mk, inl = np.zeros(12), np.zeros(36), np.zeros(N, dtype=np.uint8), np.zeros(N, dtype=np.int32)
            keep.append((X, x, K, Rt, cov, mk, inl))
            j = jobs[k]
            j.X, j.x, j.K, j.n, j.max_iteration, j.seed, j.precision, j.refine, j.huber_a = X.ctypes.data, x.ctypes.data, K.ctypes.data, N, 256, 5 + k, float("inf"), 1, 16.0
            j.Rt, j.cov, j.inlier_mask, j.inliers = Rt.ctypes.data, cov.ctypes.data, mk.ctypes.data, inl.ctypes.data
        hs = (C.c_void_p * 3)(*[c.h for c in ctxs])
        rc = lib.clc_pnp_localize_ac_batch(hs, jobs, 3)
        assert rc != CLC_OK and jobs[1].status == rc and jobs[1].n_inliers == 0
        for k in (0, 2):
            assert jobs[k].status == CLC_OK and jobs[k].n_inliers > 0
            want = gpu_ctx.pnp_acransac(scs[k]["X"], scs[k]["x"], scs[k]["K"], max_iteration=256, seed=5 + k, refine=True)
            assert np.array_equal(keep[k][3].reshape(3, 4), want["Rt"]) and np.array_equal(keep[k][6][:jobs[k].n_inliers], want["inliers"])
        # the refused job's context is still good for a solve of its own
        again = ctxs[1].pnp_acransac(scs[0]["X"], scs[0]["x"], scs[0]["K"], max_iteration=256, seed=5, refine=True)
        assert np.array_equal(again["Rt"], keep[0][3].reshape(3, 4))
        assert lib.clc_pnp_localize_ac_batch(hs, jobs, 0) == CLC_OK and lib.clc_pnp_localize_ac_batch(None, None, 2) != CLC_OK
    finally:
        for c in ctxs:
            c.close()
