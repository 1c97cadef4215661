"""clc_essential_acransac on bench.py's two-view problem (1000 correspondences, 30 % outliers): p50 over 100 solves, rounds and iterations;
under rocprofv3 --kernel-trace --stats the per-kernel averages of its launches.  usage: time_two_view.py [n_pairs_in_batch]
       time_two_view.py model F|H   the same sizes under RobustMatcher's other two models (clc_two_view_acransac; 'H' on a planar scene), singles
                                    and batches of 4 / 8"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from coloc_amd import Context
rng2 = np.random.default_rng(11)
Nc = 1000
Xs = np.stack([rng2.uniform(-5, 5, Nc), rng2.uniform(-5, 5, Nc), rng2.uniform(4, 20, Nc)], 1)
Kc = np.array([[1000.0, 0, 640], [0, 1000.0, 360], [0, 0, 1]])
ang = 0.2
Rc = np.array([[np.cos(ang), 0, -np.sin(ang)], [0, 1, 0], [np.sin(ang), 0, np.cos(ang)]])
p1 = Xs @ Kc.T; p1 = p1[:, :2] / p1[:, 2:3]
p2 = (Xs @ Rc.T + np.array([0.5, 0.1, 0.2])) @ Kc.T; p2 = p2[:, :2] / p2[:, 2:3] + rng2.normal(0, 0.5, (Nc, 2))
oi = rng2.choice(Nc, 300, replace=False)
p2[oi] = np.stack([rng2.uniform(0, 1280, 300), rng2.uniform(0, 720, 300)], 1)
if len(sys.argv) > 2 and sys.argv[1] == "model":
    from coloc_amd.abi import two_view_acransac_batch
    mdl = sys.argv[2]
    b2 = p2
    if mdl == "H":
        nrm = np.array([0.1, -0.05, 1.0]); nrm /= np.linalg.norm(nrm)
        rays = np.c_[p1, np.ones(Nc)] @ np.linalg.inv(Kc).T
        Xp = rays * (9.0 / (rays @ nrm))[:, None]
        b2 = (Xp @ Rc.T + np.array([0.5, 0.1, 0.2])) @ Kc.T; b2 = b2[:, :2] / b2[:, 2:3] + rng2.normal(0, 0.5, (Nc, 2))
        b2[oi] = p2[oi]
    ctx = Context(device=0, detector=False, matcher=False)
    tm, its = [], []
    for it in range(105):
        t1 = time.perf_counter()
        r = ctx.two_view_acransac(mdl, p1, b2, (1280, 720), max_iteration=256, seed=it + 1)
        tm.append((time.perf_counter() - t1) * 1e3); its.append(r["iterations"])
    tm = np.sort(tm[5:])
    print("two_view_acransac '%s' p50 %.3f ms  p95 %.3f  iterations median %.0f  inliers %d  threshold %.2f px" % (mdl, tm[len(tm) // 2], tm[int(len(tm) * .95)], np.median(its), len(r["inliers"]), r["error_max"]))
    ctx.close()
    for nb in (4, 8):
        ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(nb)]
        probs = [(p1, b2, None, None, (1280, 720), 100 + i) for i in range(nb)]
        tb = []
        for rep in range(60):
            t1 = time.perf_counter(); got = two_view_acransac_batch(ctxs, mdl, probs); tb.append((time.perf_counter() - t1) * 1e3)
        tb = np.sort(tb[5:])
        print("batch of %d pairs: p50 %.3f ms = %.3f per pair" % (nb, tb[len(tb) // 2], tb[len(tb) // 2] / nb))
        for c in ctxs: c.close()
    sys.exit(0)
ctx = Context(device=0, detector=False, matcher=False)
te, its, rounds = [], [], []
for it in range(105):
    t1 = time.perf_counter()
    r = ctx.essential_acransac(p1, p2, Kc, Kc, (1280, 720), max_iteration=256, seed=it + 1)
    te.append((time.perf_counter() - t1) * 1e3); its.append(r["iterations"]); rounds.append(r.get("rounds", -1))
te = np.sort(te[5:])
print("essential_acransac p50 %.3f ms  p95 %.3f  iterations median %.0f  rounds median %.0f  inliers %d" % (te[len(te) // 2], te[int(len(te) * .95)], np.median(its), np.median(rounds), len(r["inliers"])))
ctx.close()

# the same problem as a batch of n pairs (clc_essential_acransac_batch; CLC_ACR_LOCKSTEP=0: the pairs' chains interleaved instead of shared launches)
from coloc_amd.abi import essential_acransac_batch
for nb in ([int(sys.argv[1])] if len(sys.argv) > 1 else [2, 4, 8]):
    ctxs = [Context(device=0, detector=False, matcher=False) for _ in range(nb)]
    probs = [(p1, p2, Kc, Kc, (1280, 720), 100 + i) for i in range(nb)]
    tb = []
    for rep in range(60):
        t1 = time.perf_counter(); got = essential_acransac_batch(ctxs, probs); tb.append((time.perf_counter() - t1) * 1e3)
    tb = np.sort(tb[5:])
    print("batch of %d pairs (CLC_ACR_LOCKSTEP=%s): p50 %.3f ms = %.3f per pair, iterations %s" % (nb, os.environ.get("CLC_ACR_LOCKSTEP", "default"), tb[len(tb) // 2], tb[len(tb) // 2] / nb, [g["iterations"] for g in got]))
    for c in ctxs: c.close()
