"""CPU tests of the K2NN oracle (reference src/CUDAK2NN.cu:46-75): independent numpy brute force,
lane-level emulation of the reference butterfly, the exact split merge, and the
edge cases the survey lists (section 4.1)."""
import numpy as np
import pytest

import lane_emulation
import synth

_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint16)


def numpy_k2nn(Q, T, threshold):
    """Independent restatement: full distance matrix, stable argmin = lowest index on ties, second
    smallest of the multiset, 100000/200000 sentinels, uint8 threshold."""
    Q = np.asarray(Q, dtype=np.uint8).reshape(-1, 64)
    T = np.asarray(T, dtype=np.uint8).reshape(-1, 64)
    nq, nt = len(Q), len(T)
    if nt == 0:
        return np.full(nq, -1, dtype=np.int32)
    D = _POP8[Q[:, None, :] ^ T[None, :, :]].sum(-1).astype(np.int64)   # (nq, nt)
    best_i = D.argmin(1)
    best_v = D[np.arange(nq), best_i]
    if nt >= 2:
        second_v = np.partition(D, 1, axis=1)[:, 1]
    else:
        second_v = np.full(nq, 100000)
    return np.where(second_v - best_v > (threshold & 0xFF), best_i, -1).astype(np.int32)


@pytest.mark.parametrize("nq,nt,thr", [(1, 1, 40), (5, 2, 40), (37, 64, 40), (256, 300, 60), (257, 513, 40),
                                       (300, 1, 40), (64, 1000, 0), (100, 777, 255)])
def test_oracle_vs_numpy(oracle, nq, nt, thr):
    Q, T = synth.planted_descriptors(nq, nt, seed=3000 + nq + nt)
    assert np.array_equal(oracle.k2nn(Q, T, thr), numpy_k2nn(Q, T, thr))


def test_oracle_vs_lane_emulation(oracle):
    Q, T = synth.planted_descriptors(75, 130, seed=31)
    for thr in (0, 40, 60):
        assert np.array_equal(oracle.k2nn(Q, T, thr), lane_emulation.k2nn_emulated(Q, T, thr))


def test_accept_and_reject_both_exercised(oracle):
    Q, T = synth.planted_descriptors(2000, 2000, seed=3000)
    m = oracle.k2nn(Q, T, 40)
    assert 0.15 < (m >= 0).mean() < 0.6


def test_nt_zero_and_one(oracle):
    Q = synth.random_descriptors(10, seed=1)
    assert (oracle.k2nn(Q, np.zeros((0, 64), np.uint8), 40) == -1).all()
    # one train vector: second_v = 100000 sentinel -> always accepted (CUDAK2NN.cu:54,67-72)
    m, b, s = oracle.k2nn(Q, Q[:1], 40, want_dist=True)
    assert (m == 0).all() and b[0] == 0 and (s == 65535).all()


def test_duplicate_train_ties_rejected_and_lowest_index(oracle):
    T = synth.random_descriptors(50, seed=2)
    T[17] = T[3]                        # exact duplicate
    Q = T[[3]].copy()
    m, b, s = oracle.k2nn(Q, T, 0, want_dist=True)
    assert b[0] == 0 and s[0] == 0 and m[0] == -1          # tie for the minimum -> difference 0 -> reject
    # lowest index wins among equal best distances (strict '<', :68)
    T2 = synth.random_descriptors(50, seed=4)
    q = T2[[10]].copy()
    T2[30] = T2[10]
    q[0, 0] ^= 1                        # distance 1 to both 10 and 30
    m, b, s = oracle.k2nn(q, T2, 0, want_dist=True)
    assert b[0] == 1 and s[0] == 1 and m[0] == -1
    assert oracle.k2nn_split(q, T2, 0, 5)[0] == -1


def test_threshold_edges_and_truncation(oracle):
    rng = np.random.default_rng(7)
    base = rng.integers(0, 256, 64, dtype=np.uint8)

    def flipped(k):
        bits = np.unpackbits(base)
        bits[:k] ^= 1
        return np.packbits(bits)

    T = np.stack([flipped(10), flipped(51)])   # distances 10 and 51 from base
    q = base[None]
    assert oracle.k2nn(q, T, 40)[0] == 0       # 41 > 40 accept
    assert oracle.k2nn(q, T, 41)[0] == -1      # 41 > 41 reject
    assert oracle.k2nn(q, T, 256 + 40)[0] == 0   # uint8_t truncation: 296 -> 40
    assert oracle.k2nn(q, T, 256 + 41)[0] == -1


@pytest.mark.parametrize("nsplit", [1, 2, 3, 7, 64])
def test_split_merge_is_exact(oracle, nsplit):
    Q, T = synth.planted_descriptors(300, 1000, seed=77)
    T[500] = T[20]; T[999] = T[0]       # duplicates straddling partitions
    for thr in (0, 40):
        assert np.array_equal(oracle.k2nn_split(Q, T, thr, nsplit), oracle.k2nn(Q, T, thr))


def test_omp_baseline_kernels_agree(oracle):
    """The scalar-popcount loop (BASELINE.md plan), the AVX-512 VPOPCNTDQ loop (if this CPU has it) and the
    auto-selected one all equal the sequential oracle, including tails that are not multiples of 8."""
    for nt in (1, 7, 8, 9, 250, 1001):
        Q, T = synth.planted_descriptors(333, nt, seed=100 + nt)
        want = oracle.k2nn(Q, T, 40)
        for kernel in (0, 1, -1):
            m, _ = oracle.k2nn_omp(Q, T, rule=0, threshold=40, kernel=kernel)
            assert np.array_equal(m, want), (nt, kernel)


def test_omp_baseline_timed_in_region(oracle):
    """bench.py's cpu_baseline figure: the same sweep repeated inside one parallel region, timed between team barriers."""
    Q, T = synth.planted_descriptors(400, 901, seed=9)
    want = oracle.k2nn(Q, T, 40)
    for kernel in (0, 1):
        m, nthr, best = oracle.k2nn_omp_timed(Q, T, rule=0, threshold=40, kernel=kernel, reps=3)
        assert np.array_equal(m, want) and nthr >= 1 and 0.0 < best < 5.0


def test_omp_baseline_matches_k2nn_rule(oracle):
    Q, T = synth.planted_descriptors(500, 800, seed=5)
    m, nthr = oracle.k2nn_omp(Q, T, rule=0, threshold=40)
    assert nthr >= 1 and np.array_equal(m, oracle.k2nn(Q, T, 40))
    m2, _ = oracle.k2nn_omp(Q, T, rule=1, ratio=0.8)
    assert ((m2 >= 0) | (m2 == -1)).all()


def _numpy_cpumatcher(di, pi, dj, pj, ratio=0.8):
    """Definition-level DistanceRatioMatch(ratio, BRUTE_FORCE_HAMMING, I = database, J = queries) + both dedupe passes."""
    bi = np.unpackbits(di, axis=1).astype(np.int32)
    bj = np.unpackbits(dj, axis=1).astype(np.int32)
    D = bj @ (1 - bi).T + (1 - bj) @ bi.T                    # [query j, database i]
    r2 = np.float32(ratio) * np.float32(ratio)
    seen_xy, out = set(), set()
    rows = []
    for j in range(D.shape[0]):
        o = np.argsort(D[j], kind="stable")
        d1, d2 = D[j, o[0]], D[j, o[1]]
        if np.float32(d1) < r2 * np.float32(d2):
            rows.append((float(pi[o[0], 0]), float(pi[o[0], 1]), float(pj[j, 0]), float(pj[j, 1]), int(o[0]), j))
    for r in sorted(rows):
        if r[:4] in seen_xy:
            continue
        seen_xy.add(r[:4])
        out.add((r[4], r[5]))
    return out


def test_cpumatcher_pair_semantics(oracle):
    """CPUMatcher::computeMatchesPair (CPUMatcher.hpp:67-76) restated: database = first region set, queries = second,
    IndMatch(i_ = database, j_ = query), ratio 0.8^2 on integer distances, and the coordinate de-duplication."""
    rng = np.random.default_rng(9)
    Q, T = synth.planted_descriptors(300, 420, seed=77)       # Q rows are noisy copies of T rows
    di, dj = T, Q
    pi = rng.integers(0, 640, (di.shape[0], 2)).astype(np.float32)
    pj = rng.integers(0, 480, (dj.shape[0], 2)).astype(np.float32)
    # two queries at the same position matched to database rows at the same position -> one of them is dropped
    dj[1] = dj[0]; pj[1] = pj[0]
    pairs, nthr = oracle.cpumatcher_pair(di, pi, dj, pj)
    want = _numpy_cpumatcher(di, pi, dj, pj)
    assert nthr >= 1 and len(want) > 50
    assert set(map(tuple, pairs.tolist())) == want
    assert (pairs[:, 0] < di.shape[0]).all() and (pairs[:, 1] < dj.shape[0]).all()
    assert not {(int(a), 1) for a, b in pairs if b == 1} or not {(int(a), 0) for a, b in pairs if b == 0}
    # the direction matters: swapping the arguments searches the other way and swaps the roles of i_ and j_
    swapped, _ = oracle.cpumatcher_pair(dj, pj, di, pi)
    assert set(map(tuple, swapped.tolist())) == _numpy_cpumatcher(dj, pj, di, pi)
    # degenerate region sets give no match
    assert oracle.cpumatcher_pair(di[:1], pi[:1], dj, pj)[0].shape[0] == 0
    assert oracle.cpumatcher_pair(di, pi, dj[:0], pj[:0])[0].shape[0] == 0
    # both inner loops agree
    for kernel in (0, 1):
        p2, _ = oracle.cpumatcher_pair(di, pi, dj, pj, kernel=kernel)
        assert np.array_equal(p2, pairs)
