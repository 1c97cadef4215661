"""Round 5: several two-view filters at once (clc_essential_acransac_batch) and the inter-camera step behind the C ABI
(clc_inter_pose_batch; reference include/coloc/coloc.hpp:296-340, RobustMatcher.hpp:153-186, colocUtils.hpp:184-211).
The batch must give, job by job, exactly what the single-solve entry gives (same E, same inliers: the chains of launches only
interleave); the inter-camera chain must land on the destination camera's true pose and agree with a numpy statement of the same chain."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
K = np.array([[1000.0, 0, 640], [0, 1000.0, 360], [0, 0, 1]])
WH = (1280, 720)


def _rot(ax, a):
    c, s = np.cos(a), np.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]), "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[ax]


def _pair(seed, n=900, outliers=0.3, noise=0.4):
    """A world of n points, a source and a destination camera looking at it, n correspondences (30 % of the destination's replaced)."""
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-5, 5, n), rng.uniform(-5, 5, n), rng.uniform(6, 18, n)], 1)
    Rs, ts = _rot("y", rng.uniform(-0.1, 0.1)) @ _rot("x", rng.uniform(-0.05, 0.05)), rng.uniform(-0.3, 0.3, 3)
    Rd = _rot("y", rng.uniform(0.1, 0.25)) @ _rot("z", rng.uniform(-0.05, 0.05)) @ Rs
    td = ts + np.array([rng.uniform(0.6, 1.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2)])

    def proj(R, t):
        u = (X @ R.T + t) @ K.T
        return u[:, :2] / u[:, 2:3]
    x1 = proj(Rs, ts) + rng.normal(0, noise, (n, 2))
    x2 = proj(Rd, td) + rng.normal(0, noise, (n, 2))
    out = rng.choice(n, int(outliers * n), replace=False)
    x2[out] = np.stack([rng.uniform(0, WH[0], len(out)), rng.uniform(0, WH[1], len(out))], 1)
    # the global map holds 60 % of the points (in a shuffled order); the source's map matches point at them
    in_map = rng.random(n) < 0.6
    order = rng.permutation(np.nonzero(in_map)[0])
    map_X = X[order] + rng.normal(0, 0.002, (len(order), 3))
    map_index = np.full(n, -1, np.int32)
    map_index[order] = np.arange(len(order), dtype=np.int32)
    return dict(x1=x1, x2=x2, K=K, wh=WH, seed=seed, map_index=map_index, Rt_source=np.c_[Rs, ts], Rd=Rd, td=td, map_X=map_X)


def test_two_view_batch_equals_the_single_solves():
    from coloc_amd import Context
    from coloc_amd.abi import essential_acransac_batch
    ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(5)]
    try:
        pairs = [_pair(100 + i, n=300 + 250 * i) for i in range(5)]
        probs = [(p["x1"], p["x2"], K, K, WH, 7 + i) for i, p in enumerate(pairs)]
        probs[3] = (pairs[3]["x1"][:4], pairs[3]["x2"][:4], K, K, WH, 1)           # fewer correspondences than a minimal sample: no model
        got = essential_acransac_batch(ctxs, probs)
        for i, (x1, x2, _, _, _, seed) in enumerate(probs):
            one = ctxs[0].essential_acransac(x1, x2, K, K, WH, max_iteration=256, seed=seed)
            assert got[i]["status"] == 0 and got[i]["iterations"] == one["iterations"]
            assert np.array_equal(got[i]["inliers"], one["inliers"]) and got[i]["error_max"] == one["error_max"] and got[i]["min_nfa"] == one["min_nfa"]
            if one["E"] is None:
                assert got[i]["E"] is None
            else:
                assert np.array_equal(got[i]["E"], one["E"]) and np.array_equal(got[i]["F"], one["F"])
        assert got[3]["E"] is None and len(got[0]["inliers"]) > 150
    finally:
        for c in ctxs:
            c.close()


def _numpy_chain(c, p, e, common=None):
    """the chain bench_stream.py ran in numpy until round 4, on the batch's own E / inliers.  common(front correspondences) -> (global map
    index, temporary map point) pairs in the order the scale rule walks them; default: the shortcut (map_index, correspondence order)"""
    import bench_stream as bs
    inl = e["inliers"]
    x1, x2 = p["x1"], p["x2"]
    rp = bs.relative_pose_from_essential(e["E"], K, x1[inl], x2[inl])
    Rr, tr, Xtmp, front = rp
    Xtmp, x2i, qi = Xtmp[front], x2[inl][front], inl[front]
    if common is None:
        ms = p["map_index"][qi]
        com = np.nonzero(ms >= 0)[0]
        gidx = ms[com]
    else:
        gidx, com = common(qi)
    S = p["Rt_source"]
    Xg = p["map_X"][gidx]
    Xg_s = Xg @ S[:, :3].T + S[:, 3]
    ratio = np.linalg.norm(Xg_s, axis=1) / np.maximum(np.linalg.norm(Xtmp[com], axis=1), 1e-12)
    keep = np.abs(ratio / np.median(ratio) - 1.0) < 0.2
    com, Xg = com[keep], Xg[keep]
    # colocUtils.hpp:201-204: `float dist1 = (X12 - X11).norm(); float dist2 = ...; scale += dist1 / dist2` -- the norms and their ratio in float
    d1 = np.linalg.norm(Xg[1:] - Xg[:-1], axis=1).astype(np.float32)
    d2 = np.linalg.norm(Xtmp[com][1:] - Xtmp[com][:-1], axis=1).astype(np.float32)
    good = d2 > np.float32(1e-9)
    scale = float(np.sum((d1[good] / d2[good]).astype(np.float64)) / good.sum())
    Rt0 = np.c_[Rr @ S[:, :3], Rr @ S[:, 3] + scale * tr]
    Xw = (scale * Xtmp - S[:, 3]) @ S[:, :3]
    Rt_i, cov_i, rmse_i, _ = c.pnp_refine(Xw, x2i, K, Rt0)
    return dict(scale=scale, Rt=Rt_i, cov=cov_i, rmse=rmse_i, n_front=int(front.sum()), n_common=len(com), n_raw=len(gidx))


def test_inter_pose_batch_lands_on_the_destination_pose():
    from coloc_amd import Context
    from coloc_amd.abi import inter_pose_batch
    ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(4)]
    try:
        pairs = [_pair(200 + i, n=700 + 100 * i) for i in range(4)]
        mapX = [p["map_X"] for p in pairs]
        # (every pair has its own world here: one call per map; the streaming loop shares one map and makes one call)
        res = [inter_pose_batch([ctxs[i]], [pairs[i]], mapX[i])[0] for i in range(4)]
        # ... and all four pairs of ONE world in one call
        world = _pair(300, n=1000)
        probs = []
        for i in range(4):
            q = dict(world)
            q["seed"] = 50 + i
            probs.append(q)
        res_b = inter_pose_batch(ctxs, probs, world["map_X"])
        for p, r in list(zip(pairs, res)) + list(zip(probs, res_b)):
            assert r["status"] == 0 and r["stage"] == 0, r["stage"]
            assert len(r["inliers"]) > 0.55 * len(p["x1"]) and r["n_front"] > 0.9 * len(r["inliers"]) and r["n_common"] > 100
            Rt = r["Rt"]
            ang = np.degrees(np.arccos(np.clip((np.trace(Rt[:, :3] @ p["Rd"].T) - 1) / 2, -1, 1)))
            Cd, Ce = -p["Rd"].T @ p["td"], -Rt[:, :3].T @ Rt[:, 3]
            # sanity bounds only (0.4 px noise, ~1 unit baseline at 12 units of depth: the temporary map's depths are noisy and its scale --
            # the reference's rule, mean of consecutive distance ratios, colocUtils.hpp:184-211 -- is good to a few per cent of the
            # baseline); the parity statement is the comparison with the numpy chain below
            assert ang < 0.6 and np.linalg.norm(Ce - Cd) < 0.2, (ang, np.linalg.norm(Ce - Cd))
            assert 0 < r["rmse"] < 2.0 and np.all(np.linalg.eigvalsh(r["cov"]) > 0)
            # the same chain stated in numpy (bench_stream.py's until round 4) from the same E and inliers: the host arithmetic differs in
            # the order of its sums and in the decomposition of E (Jacobi here, LAPACK there), so agreement is to 1e-6, not bitwise
            ref = _numpy_chain(ctxs[0], p, r)
            assert r["n_front"] == ref["n_front"] and r["n_common"] == ref["n_common"]
            assert abs(r["scale"] / ref["scale"] - 1) < 1e-9 and np.allclose(r["Rt"], ref["Rt"], atol=1e-6) and abs(r["rmse"] - ref["rmse"]) < 1e-6
        # one call's jobs are solved like single calls: same E and inliers as clc_essential_acransac with the job's seed
        one = ctxs[0].essential_acransac(world["x1"], world["x2"], K, K, WH, max_iteration=256, seed=52)
        assert np.array_equal(res_b[2]["inliers"], one["inliers"]) and np.array_equal(res_b[2]["E"], one["E"])
        # stages that cannot complete say so: no map features -> no scale; fewer correspondences than the filter keeps -> no model
        # (uniformly random correspondences are NOT such a case: the a-contrario rule, whose alpha = e 2 D / A is not capped at 1, calls
        # "98 % of the points within 400 px of their epipolar lines" meaningful -- OpenMVG's formula, and the oracle's)
        q = dict(world); q["map_index"] = np.full(len(world["x1"]), -1, np.int32)
        assert inter_pose_batch([ctxs[0]], [q], world["map_X"])[0]["stage"] == 3
        q = dict(world); q["x1"], q["x2"], q["map_index"] = world["x1"][:9], world["x2"][:9], world["map_index"][:9]
        assert inter_pose_batch([ctxs[0]], [q], world["map_X"])[0]["stage"] == 1
        # map indices outside the map are "not a map feature", not a read past the array: with every index out of range -> no scale
        q = dict(world); q["map_index"] = np.where(world["map_index"] >= 0, world["map_index"] + 10 ** 6, -1).astype(np.int32)
        assert inter_pose_batch([ctxs[0]], [q], world["map_X"])[0]["stage"] == 3
    finally:
        for c in ctxs:
            c.close()


def test_lockstep_batches_equal_interleaved_batches_equal_single_solves():
    """Round 5: the solves of a batch may share their launches (lockstep: one launch per round for all of them, blockIdx.y = solve) instead of
    interleaving chains of their own -- the default for two-view batches of four or more, CLC_ACR_LOCKSTEP=1 / =0 forces it on / off for
    both kinds.  A chain's kernels take their batch from the chain's own device state; the shared grid and the shared sort width (the
    widest chain's) only bound them: every job -- ten two-view filters of 150 .. 2 400 correspondences (two launches per round: more
    than eight chains), a job below the minimal sample, six resection solves with refinement -- identical in both forms and to the
    single-solve entries; a third run caps a round at five iterations (CLC_ACR_BATCH_CAP: lockstep groups use 8 / 12): the
    schedule changes, no result does."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_two_view_batch as t, synth\n"
            "from coloc_amd import Context\n"
            "from coloc_amd.abi import essential_acransac_batch, pnp_localize_batch\n"
            "ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(10)]\n"
            "ns = [150, 2400, 600, 900, 1100, 300, 1700, 450, 1300, 800]\n"
            "pairs = [t._pair(500 + i, n=ns[i]) for i in range(10)]\n"
            "probs = [(p['x1'], p['x2'], t.K, t.K, t.WH, 31 + i) for i, p in enumerate(pairs)]\n"
            "probs[4] = (pairs[4]['x1'][:5], pairs[4]['x2'][:5], t.K, t.K, t.WH, 1)\n"
            "out = dict(tv=[], tv1=[], pnp=[], pnp1=[])\n"
            "def tv(r): return dict(E=None if r['E'] is None else r['E'].tolist(), inl=r['inliers'].tolist(), emax=r['error_max'], nfa=r['min_nfa'], it=r['iterations'])\n"
            "for rep in range(2):\n"
            "    got = essential_acransac_batch(ctxs, probs)\n"
            "out['tv'] = [tv(r) for r in got]\n"
            "out['tv1'] = [tv(ctxs[0].essential_acransac(x1, x2, t.K, t.K, t.WH, max_iteration=256, seed=s)) for (x1, x2, _, _, _, s) in probs]\n"
            "scenes = [synth.pnp_scene(400 + 350 * c, seed=900 + c, outlier_frac=0.3) for c in range(6)]\n"
            "pp = [(s['X'], s['x'], s['K']) for s in scenes]\n"
            "def pn(r): return dict(Rt=np.asarray(r['Rt']).tolist(), cov=np.asarray(r['cov']).tolist(), inl=np.asarray(r['inliers']).tolist(), it=r['iterations'], rmse=r['rmse'])\n"
            "for rep in range(2):\n"
            "    gp = pnp_localize_batch(ctxs[:6], pp, max_iteration=256, seeds=list(range(21, 27)), refine=True)\n"
            "out['pnp'] = [pn(r) for r in gp]\n"
            "out['pnp1'] = [pn(pnp_localize_batch([ctxs[7]], [pp[c]], max_iteration=256, seeds=[21 + c], refine=True)[0]) for c in range(6)]\n"
            "print('RESULT' + json.dumps(out))" % (os.path.dirname(here), here))
    res = {}
    for mode in ("1", "0", "cap5"):
        env = dict(os.environ)
        env.pop("CLC_ACR_BATCH_CAP", None)
        env["CLC_ACR_LOCKSTEP"] = "1" if mode == "cap5" else mode
        if mode == "cap5":
            env["CLC_ACR_BATCH_CAP"] = "5"          # rounds of at most five iterations: another schedule, the same sequential semantics
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[mode] = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT")][0][6:])
    for mode in ("1", "0", "cap5"):
        assert res[mode]["tv"] == res[mode]["tv1"], mode
        assert res[mode]["pnp"] == res[mode]["pnp1"], mode
    assert res["1"] == res["0"] and res["1"] == res["cap5"]
    assert res["1"]["tv"][4]["E"] is None and all(len(r["inl"]) > 60 for i, r in enumerate(res["1"]["tv"]) if i != 4)
    assert all(len(r["inl"]) > 200 for r in res["1"]["pnp"])


def test_interleaved_batch_straight_before_a_lockstep_batch_on_the_same_contexts():
    """a single or interleaved solve returns when its word says "done" -- the round enqueued ahead of it is still
    queued on ITS context's stream and its keeper writes the state copies in that context's workspace.  A lockstep batch that follows
    stages every solve's workspace from the FIRST context's stream: that stream must first be ordered behind each context's own.
    Six resection solves without refinement (fewer than eight: interleaved, nothing polled to completion behind the rounds) and,
    with no synchronisation in between, six two-view filters (four or more: lockstep) on the same contexts in another order, forty
    times over; every result must be the single-solve entry's."""
    from coloc_amd import Context
    from coloc_amd.abi import essential_acransac_batch, pnp_localize_batch
    ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(6)]
    ref = Context(device=0, detector=False, matcher=False)
    try:
        scenes = [synth.pnp_scene(400 + 150 * k, seed=9100 + k, outlier_frac=0.3) for k in range(6)]
        poses = [(sc["X"], sc["x"], sc["K"]) for sc in scenes]
        pairs = [_pair(300 + i, n=350 + 200 * i) for i in range(6)]
        views = [(p["x1"], p["x2"], K, K, WH, 40 + i) for i, p in enumerate(pairs)]
        want_p = [ref.pnp_acransac(X, x, Kc, max_iteration=256, seed=11 + k, refine=False) for k, (X, x, Kc) in enumerate(poses)]
        want_v = [ref.essential_acransac(x1, x2, K, K, WH, max_iteration=256, seed=s) for (x1, x2, _, _, _, s) in views]
        rng = np.random.default_rng(3)
        for rep in range(40):
            order = list(rng.permutation(6))
            got_p = pnp_localize_batch(ctxs, poses, max_iteration=256, seeds=[11 + k for k in range(6)], refine=False)
            got_v = essential_acransac_batch([ctxs[i] for i in order], views)
            for k in range(6):
                assert np.array_equal(got_p[k]["inliers"], want_p[k]["inliers"]) and np.array_equal(got_p[k]["Rt"], want_p[k]["Rt"]), (rep, k)
                assert got_p[k]["iterations"] == want_p[k]["iterations"], (rep, k)
                assert got_v[k]["iterations"] == want_v[k]["iterations"] and np.array_equal(got_v[k]["inliers"], want_v[k]["inliers"]), (rep, k)
                assert np.array_equal(got_v[k]["E"], want_v[k]["E"]) and got_v[k]["min_nfa"] == want_v[k]["min_nfa"], (rep, k)
    finally:
        for c in ctxs + [ref]:
            c.close()


def test_inter_pose_through_the_references_own_chain(oracle):
    """the common features of the temporary and the global map found the reference's way -- setupMapDatabase(inter)
    keeps the descriptor of every temporary map point's first observation (the pair's camera with the lower id, colocData.hpp:109-117),
    matchMapFeatures matches the global map's descriptors against them (K2NN, Q = map, T = temporary map, threshold 60,
    coloc.hpp:317-323) -- on the device, from the camera's descriptor block and the map's block where they lie.  Against a numpy
    statement of the same chain whose K2NN is the oracle's; and against the shortcut (the source frame's map indices): the same pose to
    the scale rule's noise.  Source id below / above the destination's: the lower camera's rows are the ones gathered."""
    import torch
    from coloc_amd import Context
    from coloc_amd.abi import inter_pose_batch
    ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(4)]
    try:
        world = _pair(410, n=1200)
        n, M = len(world["x1"]), len(world["map_X"])
        rng = np.random.default_rng(9)
        point = synth.random_descriptors(n, seed=411)                 # a descriptor per world point ...

        def noisy(rows, flips):
            d = rows.copy()
            for r in range(len(d)):
                for b in rng.choice(512, flips, replace=False):
                    d[r, b >> 3] ^= 1 << (b & 7)
            return d
        # ... seen with a few bits flipped by the source frame, by the destination frame (in ANOTHER row order) and by the map
        src_desc = noisy(point, 12)
        perm = rng.permutation(n)
        dst_desc = np.empty_like(point); dst_desc[perm] = noisy(point, 14)          # destination feature of correspondence i: row perm[i]
        in_map = np.nonzero(world["map_index"] >= 0)[0]
        map_desc = np.empty((M, 64), np.uint8)
        map_desc[world["map_index"][in_map]] = noisy(point[in_map], 10)
        d_src, d_dst, d_map = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (src_desc, dst_desc, map_desc))
        probs = []
        for i in range(4):
            q = dict(world)
            q["seed"] = 70 + i
            q.pop("map_index")
            if i % 2 == 0:     # source id < destination id: the temporary map keeps the SOURCE frame's descriptors
                q.update(d_first_desc=d_src.data_ptr(), first_feature=np.arange(n, dtype=np.int32), d_map_desc=d_map.data_ptr())
            else:              # destination id lower: the destination frame's
                q.update(d_first_desc=d_dst.data_ptr(), first_feature=perm.astype(np.int32), d_map_desc=d_map.data_ptr())
            probs.append(q)
        res = inter_pose_batch(ctxs, probs, world["map_X"])
        short = inter_pose_batch(ctxs, [dict(world, seed=70 + i) for i in range(4)], world["map_X"])
        for i, (p, r, sh) in enumerate(zip(probs, res, short)):
            assert r["status"] == 0 and r["stage"] == 0, (i, r["stage"])
            first = src_desc if i % 2 == 0 else dst_desc
            rows = p["first_feature"]

            def common(qi):
                m = oracle.k2nn(map_desc, first[rows[qi]], 60)        # Q = global map, T = the temporary map's descriptors
                g = np.nonzero(m >= 0)[0]
                return g, m[g]
            ref = _numpy_chain(ctxs[0], dict(world), r, common)
            assert r["n_front"] == ref["n_front"] and r["n_map_matches"] == ref["n_raw"] and r["n_common"] == ref["n_common"]
            assert r["n_map_matches"] > 0.5 * len(in_map) * len(r["inliers"]) / n
            assert abs(r["scale"] / ref["scale"] - 1) < 1e-9 and np.allclose(r["Rt"], ref["Rt"], atol=1e-6) and abs(r["rmse"] - ref["rmse"]) < 1e-6
            # the two sources of common features walk (nearly) the same features in different orders: the same filter result, poses that
            # agree to the noise of the scale rule
            assert np.array_equal(r["inliers"], sh["inliers"]) and r["n_front"] == sh["n_front"]
            assert abs(r["scale"] / sh["scale"] - 1) < 0.02
            Cd = -p["Rd"].T @ p["td"]
            for rr in (r, sh):
                Ce = -rr["Rt"][:, :3].T @ rr["Rt"][:, 3]
                assert np.linalg.norm(Ce - Cd) < 0.2
        # all three pointers or none
        bad = dict(probs[0]); bad.pop("first_feature")
        with pytest.raises(Exception):
            inter_pose_batch([ctxs[0]], [bad], world["map_X"])
    finally:
        for c in ctxs:
            c.close()
