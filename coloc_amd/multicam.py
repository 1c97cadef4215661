"""One camera per GPU: describe locally, all-gather the 512-bit descriptors over xGMI (RCCL), then
every rank matches its share of the all-pairs sweep.

The reference has no multi-GPU path: multi-camera matching is a serial loop over
`Utils::handlePairs(n)` = openMVG exhaustivePairs (include/coloc/colocUtils.hpp:58-61) inside one
process (include/coloc/GPUMatcher.hpp:143-155).  This module keeps that loop's RESULT (the same
`IndMatches` per (first < second) pair, Q = descriptors of `first`, T = of `second`,
GPUMatcher.hpp:165-172) and re-cuts the work: after the all-gather every rank holds every camera's
descriptors, so the flattened (pair, query-block) space is dealt out in equal contiguous shares --
no further data-path collective is needed.

The planning functions are pure Python (tested on CPU); the collective is torch.distributed
(backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) -- plumbing only.
"""
from dataclasses import dataclass
from typing import List, Tuple

# Default grain of a share, in queries.  A K2NN sweep workgroup covers clc_k2nn_queries_per_block() queries (256 for the
# matrix formulation, 128 for the popcount one, coloc_amd/csrc/k2nn.hip); cutting shares on a multiple of it means no
# workgroup straddles two ranks.  Callers that hold a context pass grain=ctx.k2nn_queries_per_block; the default is a
# multiple of both (tests/test_abi.py checks that).
QBLOCK = 256


def exhaustive_pairs(n: int) -> List[Tuple[int, int]]:
    """All (i, j), i < j, lexicographic -- openMVG exhaustivePairs as used by colocUtils.hpp:58-61."""
    return [(i, j) for i in range(n) for j in range(i + 1, n)]


@dataclass(frozen=True)
class Job:
    pair: Tuple[int, int]   # (first, second) camera ids
    q_begin: int            # first query row inside camera `first`
    nq: int
    out_offset: int         # first int32 of this job's results inside the rank-local result buffer


def shard_pairs(counts: List[int], world: int, rank: int, qblock: int = QBLOCK, grain: int = 0) -> List[Job]:
    """Jobs of `rank`: a contiguous share of the (pair, query-block) sequence, split as evenly as the
    block grain allows.  counts[c] = number of descriptors of camera c.  grain (if given) overrides qblock."""
    if grain > 0:
        qblock = grain
    pairs = exhaustive_pairs(len(counts))
    blocks = []                                    # (pair index, block index) flattened
    for p, (i, j) in enumerate(pairs):
        if counts[i] == 0 or counts[j] == 0:
            continue                               # an empty side yields no matches (GPUMatcher.hpp:150)
        blocks.extend((p, b) for b in range((counts[i] + qblock - 1) // qblock))
    total = len(blocks)
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    jobs: List[Job] = []
    out = 0
    k = lo
    while k < hi:
        p, b0 = blocks[k]
        b1 = b0
        while k + 1 < hi and blocks[k + 1][0] == p:
            k += 1
            b1 = blocks[k][1]
        i, j = pairs[p]
        q_begin = b0 * qblock
        nq = min(counts[i], (b1 + 1) * qblock) - q_begin
        jobs.append(Job(pairs[p], q_begin, nq, out))
        out += nq
        k += 1
    return jobs


def jobs_to_abi(jobs: List[Job], counts: List[int], cap: int, threshold: int):
    """clc_match_job tuples over the gathered arena laid out [camera][cap rows][64 B]."""
    return [(j.pair[0] * cap + j.q_begin, j.nq, j.pair[1] * cap, counts[j.pair[1]], j.out_offset, threshold)
            for j in jobs]


def all_gather_descriptors(my_desc, my_count: int, world: int, group=None):
    """Fixed-capacity all-gather: `my_desc` is this rank's [cap, 64] uint8 tensor (rows >= my_count are
    padding).  Returns ([world, cap, 64] tensor, list of counts).  The count rides in a second tiny
    all-gather; at 10k descriptors the payload is 640 KB per rank -- latency-bound on xGMI."""
    import torch
    import torch.distributed as dist
    cap = my_desc.shape[0]
    gathered = torch.empty((world, cap, 64), dtype=torch.uint8, device=my_desc.device)
    cnt = torch.tensor([my_count], dtype=torch.int32, device=my_desc.device)
    cnts = torch.empty((world,), dtype=torch.int32, device=my_desc.device)
    if world == 1:
        gathered[0].copy_(my_desc)
        cnts.copy_(cnt)
    else:
        dist.all_gather_into_tensor(gathered.view(-1), my_desc.reshape(-1), group=group)
        dist.all_gather_into_tensor(cnts, cnt, group=group)
    return gathered, [int(v) for v in cnts.tolist()]


def assemble_pairwise(results, world_jobs: List[List[Job]], counts: List[int]):
    """Rebuild {pair: int32[counts[first]]} (the single-GPU loop's per-pair match arrays) from every
    rank's result buffer.  results[r] is rank r's int32 array (host)."""
    import numpy as np
    out = {}
    for r, jobs in enumerate(world_jobs):
        for j in jobs:
            arr = out.setdefault(j.pair, np.full(counts[j.pair[0]], -1, dtype=np.int32))
            arr[j.q_begin:j.q_begin + j.nq] = results[r][j.out_offset:j.out_offset + j.nq]
    return out


def ind_matches(pair_result):
    """IndMatch list of one pair: only accepted queries, ascending query index, (i_ = query idx,
    j_ = train idx) -- GPUMatcher.hpp:215-220."""
    return [(int(q), int(t)) for q, t in enumerate(pair_result) if t != -1]
