"""One-camera-per-GPU orchestration (coloc_amd/multicam.py) on CPU: the pair/query-block sharding
covers the all-pairs loop of GPUMatcher::computeMatches (include/coloc/GPUMatcher.hpp:143-155)
exactly once, and a world_size-2 gloo run (all-gather of fixed-capacity descriptor blocks, per-rank
job lists, reassembly) reproduces the single-process loop bit for bit.  The device compute is
injected by the test (the CPU oracle) -- the product path has no CPU backend."""
import os
import socket
import sys

import numpy as np
import pytest

import synth
from coloc_amd import multicam

HERE = os.path.dirname(os.path.abspath(__file__))


def test_exhaustive_pairs_order():
    assert multicam.exhaustive_pairs(4) == [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    assert multicam.exhaustive_pairs(1) == []


@pytest.mark.parametrize("counts,world", [([10000, 10000], 1), ([10000, 10000], 2), ([10000] * 4, 4), ([10000] * 8, 8),
                                          ([700, 0, 1300, 512, 513], 3), ([5, 7], 4), ([100] * 8, 3)])
def test_sharding_is_an_exact_partition(counts, world):
    seen = {}
    loads = []
    for r in range(world):
        jobs = multicam.shard_pairs(counts, world, r)
        off = 0
        for j in jobs:
            assert j.out_offset == off
            off += j.nq
            assert j.nq > 0 and j.q_begin % multicam.QBLOCK == 0
            cov = seen.setdefault(j.pair, np.zeros(counts[j.pair[0]], dtype=np.int32))
            cov[j.q_begin:j.q_begin + j.nq] += 1
        loads.append(sum(j.nq * counts[j.pair[1]] for j in jobs))
    for (i, k) in multicam.exhaustive_pairs(len(counts)):
        if counts[i] == 0 or counts[k] == 0:
            assert (i, k) not in seen
        else:
            assert (seen[(i, k)] == 1).all()
    if min(counts) == max(counts) and counts[0] >= 10000:
        assert max(loads) - min(loads) <= multicam.QBLOCK * counts[0]      # balanced to one block


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, counts, cap, thr, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = oracle_lib.Oracle()
    # this rank's camera: descriptors padded to the fixed capacity
    d = synth.random_descriptors(counts[rank], seed=3000 + rank)
    if rank > 0:   # plant matches against camera 0 so results are non-trivial
        src = synth.random_descriptors(counts[0], seed=3000)
        k = min(len(d), len(src)) // 2
        d[:k] = src[:k]
        d[:k, 0] ^= np.arange(k, dtype=np.uint8)
    mine = torch.zeros((cap, 64), dtype=torch.uint8)
    mine[:counts[rank]] = torch.from_numpy(d)
    gathered, cnts = multicam.all_gather_descriptors(mine, counts[rank], world)
    assert cnts == counts
    arena = gathered.numpy().reshape(world * cap, 64)
    jobs = multicam.shard_pairs(cnts, world, rank)
    res = np.full(max(1, sum(j.nq for j in jobs)), -7, dtype=np.int32)
    for (q_off, nq, t_off, nt, out_off, threshold) in multicam.jobs_to_abi(jobs, cnts, cap, thr):
        res[out_off:out_off + nq] = orc.k2nn(arena[q_off:q_off + nq], arena[t_off:t_off + nt], threshold)
    np.save(os.path.join(out_dir, "res%d.npy" % rank), res)
    np.save(os.path.join(out_dir, "desc%d.npy" % rank), d)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[1500, 1100], [600, 0]])
def test_world2_gloo_matches_single_process_loop(tmp_path, oracle, counts):
    import torch.multiprocessing as mp
    world, cap, thr = 2, 2048, 40
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, counts, cap, thr, str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    descs = [np.load(tmp_path / ("desc%d.npy" % r)) for r in range(world)]
    results = [np.load(tmp_path / ("res%d.npy" % r)) for r in range(world)]
    world_jobs = [multicam.shard_pairs(counts, world, r) for r in range(world)]
    got = multicam.assemble_pairwise(results, world_jobs, counts)
    # the reference's loop: for every (first < second): Q = regions[first], T = regions[second], thr 40
    for (i, j) in multicam.exhaustive_pairs(world):
        if counts[i] == 0 or counts[j] == 0:
            assert (i, j) not in got
            continue
        want = oracle.k2nn(descs[i], descs[j], thr)
        assert np.array_equal(got[(i, j)], want)
        im = multicam.ind_matches(got[(i, j)])
        assert im == [(int(q), int(t)) for q, t in enumerate(want) if t != -1]
        assert len(im) > 100


def test_c_planner_equals_python_planner():
    """clc_mc_plan (the planner a C++ host uses, coloc_amd/csrc/multicam.hip) deals out exactly the shares of
    multicam.shard_pairs -- pure host arithmetic, callable without a GPU."""
    from coloc_amd import mc_plan
    rng = np.random.default_rng(3)
    cases = [([10000] * 8, 8), ([10000, 10000], 2), ([5, 0, 700, 1300], 3), ([0, 0], 2), ([1], 1), ([257, 256, 255, 1, 4000], 4)]
    for _ in range(40):
        n = int(rng.integers(1, 9))
        cases.append(([int(c) for c in rng.integers(0, 12000, n) * (rng.random(n) > 0.15)], int(rng.integers(1, 9))))
    for counts, world in cases:
        for grain in (128, 256):
            for rank in range(world):
                want = [(j.pair[0], j.pair[1], j.q_begin, j.nq, j.out_offset) for j in multicam.shard_pairs(counts, world, rank, grain=grain)]
                assert mc_plan(counts, world, rank, grain) == want, (counts, world, rank, grain)
