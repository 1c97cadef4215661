"""Study (CPU only, round 6; result in profiles/r06_round_count_study.txt): how many rounds would an a-contrario pose solve need if
speculated iterations stayed valid across an index-set switch?  With a membership-based (rejection) sampler -- draw data indices, keep
those in the index set -- an iteration's sample is the same under the old and the new set whenever every draw's membership is, so a
round would only have to end at the first iteration whose sample really changes.  Sequential semantics unchanged; this counts rounds
for the scheme in the tree (a round ends at every switch), for one continuation per round and for unlimited continuation.
usage: python tests/soak/sim_round_count.py"""
import sys, math
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import synth, p3p_host, oracle_lib
orc = oracle_lib.Oracle()

def mix(z):
    z = (z + 0x9E3779B97F4A7C15) & (2**64 - 1)
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
    return z ^ (z >> 31)

def draws(seed, it, n, member, m):
    """rejection sampler: returns (sample ids, list of (id, accepted))"""
    s = mix(seed ^ mix(it + 1))
    ids, log = [], []
    while len(ids) < m:
        s = mix(s)
        p = ((s >> 32) * n) >> 32
        ok = member[p] and p not in ids
        log.append((p, ok))
        if ok: ids.append(p)
    return ids, log

def run(sc, seed, max_it=256, m=3):
    X, x, K = sc["X"], sc["x"], sc["K"]
    n = len(X)
    logal0, mult = math.log10(math.pi), 1.0
    norm = 1.0 / K[0, 0]
    member = np.ones(n, bool)
    reserve = max_it // 10; n_iter = max_it - reserve
    min_nfa = float("inf"); n_inl = 0; inl = None
    hist = []   # per iteration: membership version id at sampling time
    sets = [member.copy()]
    set_at = []
    it = 0
    events = []
    while it < n_iter:
        ids, _ = draws(seed, it, n, sets[-1], m)
        set_at.append(len(sets) - 1)
        poses = p3p_host.sample_poses(X, x, K, ids)
        better = False
        for P in poses:
            if np.isnan(P).any(): continue
            e = orc.pnp_residuals(P.reshape(1, 12), X, x, K)[0] * norm * norm
            v, k = orc.acr_best_nfa(e, m, 4, logal0, mult)
            if v < min_nfa:
                min_nfa = v; better = True; n_inl = k
                inl = np.argsort(e, kind="stable")[:k]
        if (better and min_nfa < 0) or (it + 1 == n_iter and reserve):
            if n_inl == 0:
                n_iter += 1; reserve -= 1
            else:
                mm = np.zeros(n, bool); mm[inl] = True
                sets.append(mm); events.append(it)
                if reserve:
                    n_iter = it + 1 + reserve; reserve = 0
        it += 1
    total = it
    # rounds, current scheme: a round ends at each event; batch sizes 32, 64, 128 while no event, then the rest
    def rounds_current():
        r, s, grow, switched = 0, 0, 32, False
        ev = set(events)
        while s < total:
            B = (total - s) if switched else min(grow, total - s)
            end = s + B
            for t in range(s, s + B):
                if t in ev: end = t + 1; switched = True; break
            else:
                grow = min(grow * 2, 128)
            s = end; r += 1
        return r
    def rounds_spec(max_switches):
        r, s, grow, switched = 0, 0, 32, False
        ev = set(events)
        while s < total:
            B = (total - s) if switched else min(grow, total - s)
            start_set = sets[set_at[s]]
            end = s + B
            nsw = 0
            for t in range(s, s + B):
                cur = sets[set_at[t]]
                if cur is not start_set:
                    # valid iff the rejection sampler takes the same decisions under both sets
                    _, log = draws(seed, t, n, start_set, m)
                    if any(bool(cur[p]) != bool(start_set[p]) for p, _ in log):
                        end = t; break
                if t in ev:
                    switched = True; nsw += 1
                    if nsw >= max_switches: end = t + 1; break
            else:
                if not switched: grow = min(grow * 2, 128)
            if end == s: end = s + 1   # (cannot happen: the first iteration of a round is sampled under the right set)
            s = end; r += 1
        return r
    return total, len(events), rounds_current(), rounds_spec(2), rounds_spec(99)

res = []
for seed in range(1, 21):
    sc = synth.pnp_scene(1000, seed=4000 + seed, outlier_frac=0.3)
    res.append(run(sc, seed))
res = np.array(res)
print("iterations, events, rounds now, rounds with one continuation, rounds with full continuation")
print(res)
print("means", res.mean(0))
