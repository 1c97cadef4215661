"""The per-rank compute of the one-camera-per-GPU path (coloc_amd/multicam.py job lists ->
clc_match_jobs_dev over the gathered descriptor arena), executed for EVERY rank of a virtual world on
the one GPU of the test box and reassembled: must equal the single-GPU all-pairs loop
(GPUMatcher.hpp:143-155) and the oracle, bit for bit.  The collective itself is covered by the gloo
test (tests/test_multicam.py); an N > 1 RCCL run is the driver's."""
import numpy as np
import pytest

import synth
from coloc_amd import multicam

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,counts", [(1, [3000, 2500]), (2, [3000, 2500]), (4, [2000, 1500, 2500, 1800]),
                                          (8, [1200] * 8), (3, [700, 0, 1300, 512, 513]),
                                          (1, [600] * 8)])   # 28 jobs in one call: two launch chunks of 16
def test_every_ranks_jobs_reassemble_to_the_all_pairs_loop(oracle, world, counts):
    import torch
    from coloc_amd import Context
    cap = max(max(counts), 1)
    ctx = Context(device=0, width=160, height=120, maxkp=cap, detector=False)
    base = synth.random_descriptors(cap, seed=1)
    descs = []
    for c, n in enumerate(counts):
        d = synth.random_descriptors(n, seed=3000 + c)
        k = n // 2
        d[:k] = base[:k]
        d[:k, c % 64] ^= (np.arange(k) % 251).astype(np.uint8)      # near-duplicates across cameras
        descs.append(d)
    arena_h = np.zeros((len(counts), cap, 64), np.uint8)
    for c, d in enumerate(descs):
        arena_h[c, :len(d)] = d
    arena = torch.from_numpy(arena_h).cuda()
    st = torch.cuda.Stream()
    results, world_jobs = [], []
    for r in range(world):
        jobs = multicam.shard_pairs(counts, world, r)
        world_jobs.append(jobs)
        out = torch.full((max(1, sum(j.nq for j in jobs)),), -9, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
        with torch.cuda.stream(st):
            ctx.match_jobs_dev(arena.data_ptr(), multicam.jobs_to_abi(jobs, counts, cap, 40), out.data_ptr(), st.cuda_stream)
        st.synchronize()
        results.append(out.cpu().numpy())
    got = multicam.assemble_pairwise(results, world_jobs, counts)
    pairs = [(i, j) for (i, j) in multicam.exhaustive_pairs(len(counts)) if counts[i] and counts[j]]
    assert sorted(got) == pairs
    loop = ctx.match_pairs(descs, pairs, 40)                     # the single-GPU loop through the host ABI
    for p, m in zip(pairs, loop):
        want = oracle.k2nn(descs[p[0]], descs[p[1]], 40)
        assert np.array_equal(got[p], want) and np.array_equal(m, want)
        assert (want >= 0).sum() > 50
    ctx.close()


@pytest.mark.parametrize("world,counts", [(2, [3000, 2500]), (4, [2000, 1500, 2500, 1800]), (8, [1200] * 8), (3, [700, 0, 1300])])
def test_c_entry_points_every_rank_of_a_virtual_world(oracle, world, counts):
    """The same reassembly through the C multi-camera entry (clc_mc_create / gather / match, coloc_amd/csrc/multicam.hip):
    one rehearsal handle per rank on the one GPU, the other ranks' blocks filed with clc_mc_virtual_put in place of the
    RCCL all-gather / peer copies (which need one GPU per rank and run on the driver's multi-GPU node)."""
    import torch
    from coloc_amd import Context, MultiCam
    cap = max(counts)
    ctx = Context(device=0, width=160, height=120, maxkp=cap, detector=False)
    descs = [synth.random_descriptors(n, seed=3100 + c) for c, n in enumerate(counts)]
    for c in range(1, world):
        k = min(len(descs[0]), len(descs[c])) // 2
        descs[c][:k] = descs[0][:k]
        descs[c][:k, c] ^= 0x5A
    dev = [torch.from_numpy(np.ascontiguousarray(np.concatenate([d, np.zeros((cap - len(d), 64), np.uint8)]))).cuda() for d in descs]
    results, shares_all = [], []
    for r in range(world):
        mc = MultiCam(ctx, world=world, rank=r, maxkp=cap)            # no id: rehearsal handle
        for o in range(world):
            if o != r:
                mc.virtual_put(o, dev[o].data_ptr(), counts[o])
        got_counts = mc.gather_dev(dev[r].data_ptr(), counts[r], mode=r % 2)
        assert got_counts == counts
        out = torch.full((max(1, cap * world),), -9, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
        shares = mc.match_dev(40, out.data_ptr(), cap * world)
        ctx.sync()
        assert shares == [(j.pair[0], j.pair[1], j.q_begin, j.nq, j.out_offset)
                          for j in multicam.shard_pairs(counts, world, r, grain=ctx.k2nn_queries_per_block)]
        results.append(out.cpu().numpy()); shares_all.append(shares)
        mc.close()
    for (i, j) in multicam.exhaustive_pairs(world):
        if not counts[i] or not counts[j]:
            continue
        got = np.full(counts[i], -7, np.int32)
        for r in range(world):
            for (a, b, q0, nq, off) in shares_all[r]:
                if (a, b) == (i, j):
                    got[q0:q0 + nq] = results[r][off:off + nq]
        assert np.array_equal(got, oracle.k2nn(descs[i], descs[j], 40))
    ctx.close()


@pytest.mark.parametrize("world,counts_by_step", [
    (2, [[3000, 2500], [2800, 3000], [100, 2900]]),
    (4, [[2000, 1500, 2500, 1800], [2500, 2500, 0, 2100], [1999, 2001, 2003, 1]]),
    (3, [[700, 0, 1300], [1300, 700, 64]]),
])
def test_enqueue_only_steps_with_different_descriptors_every_step(oracle, world, counts_by_step):
    """The streaming form of the step (clc_mc_gather_enqueue_dev + clc_mc_match_enqueue_dev): no host synchronisation between
    the exchange and the sweep, shares cut on the block capacity, the sweep reads the gathered counts on the device.  One
    rehearsal handle per rank lives across the steps; every step brings DIFFERENT descriptors and counts, and the next step's
    blocks are filed (on a second stream, as a peer's copy would arrive) while the sweep of the current step is still enqueued:
    the arena is double-buffered by step parity, so every step's result must still be the serial loop's."""
    import torch
    from coloc_amd import Context, MultiCam
    cap = max(max(c) for c in counts_by_step)
    ctx = Context(device=0, width=160, height=120, maxkp=cap, detector=False)
    st_sweep, st_copy = torch.cuda.Stream(), torch.cuda.Stream()
    steps = []
    for s, counts in enumerate(counts_by_step):
        descs = [synth.random_descriptors(n, seed=4000 + 17 * s + c) for c, n in enumerate(counts)]
        for c in range(1, world):
            k = min(len(descs[0]), len(descs[c])) // 2
            descs[c][:k] = descs[0][:k]
            if k:
                descs[c][:k, (c + s) % 64] ^= 0x3C
        dev = [torch.from_numpy(np.ascontiguousarray(np.concatenate([d, np.full((cap - len(d), 64), 0xA5, np.uint8)]))).cuda() for d in descs]
        d_cnt = [torch.tensor([n], dtype=torch.int32, device="cuda") for n in counts]
        steps.append((counts, descs, dev, d_cnt))
    torch.cuda.synchronize()
    for r in range(world):
        mc = MultiCam(ctx, world=world, rank=r, maxkp=cap)
        outs = [torch.full((cap * world,), -9, dtype=torch.int32, device="cuda") for _ in steps]
        torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
        shares_by_step = []

        def put_others(step):
            counts, _, dev, _ = steps[step]
            for o in range(world):
                if o != r:
                    mc.virtual_put(o, dev[o].data_ptr(), counts[o], stream=st_copy.cuda_stream)

        put_others(0)
        for s, (counts, descs, dev, d_cnt) in enumerate(steps):
            # own block: count from the host on even steps, read on the device (the detector's counter) on odd ones
            if s % 2 == 0:
                mc.gather_enqueue_dev(dev[r].data_ptr(), counts[r], mode=s % 2, stream=st_sweep.cuda_stream)
            else:
                mc.gather_enqueue_dev(dev[r].data_ptr(), 0, mode=s % 2, stream=st_sweep.cuda_stream, d_my_count=d_cnt[r].data_ptr())
            shares_by_step.append(mc.match_enqueue_dev(40, outs[s].data_ptr(), cap * world, stream=st_sweep.cuda_stream))
            if s + 1 < len(steps):
                put_others(s + 1)          # lands in the OTHER buffer while this step's sweep may still be running
        assert mc.counts(stream=st_sweep.cuda_stream) == steps[-1][0]
        torch.cuda.synchronize()
        want_shares = [(j.pair[0], j.pair[1], j.q_begin, j.nq, j.out_offset)
                       for j in multicam.shard_pairs([cap] * world, world, r, grain=ctx.k2nn_queries_per_block)]
        for s, (counts, descs, dev, d_cnt) in enumerate(steps):
            assert shares_by_step[s] == want_shares          # the capacity plan: the same every step
            res = outs[s].cpu().numpy()
            for (a, b, q0, nq, off) in shares_by_step[s]:
                want = oracle.k2nn(descs[a], descs[b], 40) if counts[a] and counts[b] else np.full(counts[a], -1, np.int32)
                valid = max(0, min(nq, counts[a] - q0))
                assert np.array_equal(res[off:off + valid], want[q0:q0 + valid]), (r, s, a, b)
                assert (res[off + valid:off + nq] == -1).all()       # planned rows past the camera's count
        mc.close()
    ctx.close()


def test_sync_steps_alternate_buffers(oracle):
    """clc_mc_gather_dev / clc_mc_match_dev (counts on the host, exact shares) over three steps of different data: the arena
    pointer alternates between the handle's two buffers and every step's result is the oracle's."""
    import torch
    from coloc_amd import Context, MultiCam
    world, cap = 2, 1500
    ctx = Context(device=0, width=160, height=120, maxkp=cap, detector=False)
    mc = MultiCam(ctx, world=world, rank=1, maxkp=cap)
    seen = []
    for s, counts in enumerate([[1500, 1200], [900, 1500], [1500, 1500]]):
        descs = [synth.random_descriptors(n, seed=5000 + 3 * s + c) for c, n in enumerate(counts)]
        k = min(counts) // 2
        descs[1][:k] = descs[0][:k]
        descs[1][:k, s] ^= 0x11
        dev = [torch.from_numpy(np.ascontiguousarray(np.concatenate([d, np.zeros((cap - len(d), 64), np.uint8)]))).cuda() for d in descs]
        mc.virtual_put(0, dev[0].data_ptr(), counts[0])
        assert mc.gather_dev(dev[1].data_ptr(), counts[1], mode=1) == counts
        seen.append(mc.arena())
        out = torch.full((cap * world,), -9, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
        shares = mc.match_dev(40, out.data_ptr(), cap * world)
        ctx.sync()
        want = oracle.k2nn(descs[0], descs[1], 40)
        for (a, b, q0, nq, off) in shares:
            assert (a, b) == (0, 1) and np.array_equal(out.cpu().numpy()[off:off + nq], want[q0:q0 + nq])
    assert seen[0] == seen[2] and seen[0] != seen[1]
    mc.close()
    ctx.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_one_rank_communicator_drives_the_rccl_abi(oracle, mode):
    """clc_mc_create(world = 1, WITH an id): the run-time-resolved RCCL ABI of multicam.hip (ncclGetUniqueId, ncclCommInitRank with the
    locally declared 128-byte id passed by value, ncclAllGather with ncclUint8 == 1, ncclCommDestroy) against the box's own librccl, on
    one GPU.  mode 0 = the RCCL all-gather of block + count, mode 1 = the peer-copy fan-out fenced by the counts' all-gather.  The
    gathered arena must hold the rank's rows and count, over three enqueue-only steps of different data (both buffers), and a sweep of
    the gathered block must be the oracle's."""
    import torch
    from coloc_amd import Context, MultiCam
    cap = 2048
    ctx = Context(device=0, width=160, height=120, maxkp=cap, detector=False)
    uid = MultiCam.unique_id()
    assert len(uid) == 128 and any(uid)
    mc = MultiCam(ctx, world=1, rank=0, maxkp=cap, unique_id=uid)
    other = synth.random_descriptors(1500, seed=6100)
    d_other = torch.from_numpy(other).cuda()
    st = torch.cuda.Stream()
    for s, n in enumerate([2048, 1111, 1]):
        desc = synth.random_descriptors(n, seed=6000 + s)
        k = min(n, 700)
        desc[:k] = other[:k]
        desc[:k, s] ^= 0x81
        dev = torch.from_numpy(np.ascontiguousarray(np.concatenate([desc, np.full((cap - n, 64), 0x5A, np.uint8)]))).cuda()
        d_cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
        if s == 1:
            mc.gather_enqueue_dev(dev.data_ptr(), 0, mode=mode, stream=st.cuda_stream, d_my_count=d_cnt.data_ptr())
        else:
            mc.gather_enqueue_dev(dev.data_ptr(), n, mode=mode, stream=st.cuda_stream)
        assert mc.match_enqueue_dev(40, 0, 0, stream=st.cuda_stream) == []       # one camera: no pair, nothing to sweep
        assert mc.counts(stream=st.cuda_stream) == [n]
        arena = mc.arena()
        out = torch.full((cap,), -9, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
        ctx.match_2nn_dev(arena, n, d_other.data_ptr(), len(other), 40, out.data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy()[:n], oracle.k2nn(desc, other, 40))
    mc.close()
    ctx.close()


def test_host_counts_of_four_steps_enqueued_behind_a_backed_up_stream(oracle):
    """The enqueue-only gather takes the host's count BY VALUE (a one-thread launch): four steps with four different counts are
    enqueued, without any synchronisation, behind a sweep stream that is deliberately kept busy -- the host is four steps ahead of
    the device.  (Round 3 staged the count in one pinned word per buffer: steps k and k + 2 share it, and the later count replaced the
    earlier one before its copy had executed.)  The other rank's block is the same for every step, so it can sit in both buffers."""
    import torch
    from coloc_amd import Context, MultiCam
    world, cap, r = 2, 2000, 0
    ctx = Context(device=0, width=160, height=120, maxkp=cap, detector=False)
    mine = synth.random_descriptors(cap, seed=7000)
    theirs = synth.random_descriptors(1700, seed=7001)
    theirs[:900] = mine[:900]
    theirs[:900, 5] ^= 0x42
    d_mine = torch.from_numpy(mine).cuda()
    d_theirs = torch.from_numpy(np.ascontiguousarray(np.concatenate([theirs, np.zeros((cap - len(theirs), 64), np.uint8)]))).cuda()
    mc = MultiCam(ctx, world=world, rank=r, maxkp=cap)
    st = torch.cuda.Stream()
    # the other rank's block into both buffers: put -> gather (flips the buffer) -> put
    mc.virtual_put(1, d_theirs.data_ptr(), len(theirs), stream=st.cuda_stream)
    mc.gather_enqueue_dev(d_mine.data_ptr(), 1, mode=0, stream=st.cuda_stream)
    mc.virtual_put(1, d_theirs.data_ptr(), len(theirs), stream=st.cuda_stream)
    torch.cuda.synchronize()
    counts = [2000, 300, 1234, 64]
    outs = [torch.full((cap * world,), -9, dtype=torch.int32, device="cuda") for _ in counts]
    torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
    with torch.cuda.stream(st):
        torch.cuda._sleep(int(2.0e8))                  # ~0.1 s of device time in front of everything that follows
    shares = []
    for s, n in enumerate(counts):
        mc.gather_enqueue_dev(d_mine.data_ptr(), n, mode=s % 2, stream=st.cuda_stream)
        shares.append(mc.match_enqueue_dev(40, outs[s].data_ptr(), cap * world, stream=st.cuda_stream))
    assert not st.query()                              # the host really was ahead: nothing of the four steps had finished
    torch.cuda.synchronize()
    for s, n in enumerate(counts):
        want = oracle.k2nn(mine[:n], theirs, 40)
        res = outs[s].cpu().numpy()
        for (a, b, q0, nq, off) in shares[s]:
            assert (a, b) == (0, 1)
            valid = max(0, min(nq, n - q0))
            assert np.array_equal(res[off:off + valid], want[q0:q0 + valid]), (s, n)
            assert (res[off + valid:off + nq] == -1).all()
    mc.close()
    ctx.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_config3_at_size_eight_cameras_of_ten_thousand_keypoints(oracle, overlap):
    """BASELINE config[3] AT SIZE on one GPU: 8 cameras x 10 000 descriptors, all 28 (first < second) pairs = 2.8e9
    comparisons, every rank of the virtual world through clc_mc_gather_enqueue_dev + clc_mc_match_enqueue_dev (rehearsal handles: the
    other ranks' blocks filed in place of the RCCL all-gather), over THREE steps of different descriptors so that every arena buffer is
    used and re-used -- one-stream steps and OVERLAPPED steps (clc_mc_set_overlap: step k + 1's exchange on one stream beside step k's
    sweep on another, three buffers, two event chains).  Reassembled over the ranks, every pair of every step must be the serial
    all-pairs loop's result (GPUMatcher.hpp:143-155) as the OpenMP oracle computes it (checked against the scalar oracle on one pair)."""
    import torch
    from coloc_amd import Context, MultiCam
    world, cap, nsteps = 8, 10000, 3
    ctx = Context(device=0, width=160, height=120, maxkp=cap, detector=False)
    base = synth.random_descriptors(cap, seed=9001)
    steps = []
    for s in range(nsteps):
        descs = []
        for c in range(world):
            d = synth.random_descriptors(cap, seed=9100 + 31 * s + c)
            k = 3000 + 500 * ((c + s) % 4)
            d[:k] = base[:k]
            d[:k, (c + 3 * s) % 64] ^= (np.arange(k) % 251).astype(np.uint8)        # near-duplicates across the cameras
            descs.append(d)
        steps.append((descs, [torch.from_numpy(d).cuda() for d in descs]))
    want = []
    for descs, _ in steps:
        w = {}
        for (i, j) in multicam.exhaustive_pairs(world):
            w[(i, j)], _ = oracle.k2nn_omp(descs[i], descs[j], rule=0, threshold=40)
        want.append(w)
    assert np.array_equal(want[0][(2, 5)], oracle.k2nn(steps[0][0][2], steps[0][0][5], 40))     # the OpenMP sweep pinned by the scalar one
    assert sum(int((m >= 0).sum()) for m in want[0].values()) > 28 * 1500
    st_x, st_s = torch.cuda.Stream(), torch.cuda.Stream()
    sweep_stream = st_s if overlap else st_x
    got = [dict() for _ in range(nsteps)]
    for r in range(world):
        mc = MultiCam(ctx, world=world, rank=r, maxkp=cap)
        if overlap:
            mc.set_overlap(True)
        assert mc.comm_info() == (0, -1)                              # a rehearsal handle has no communicator
        outs = [torch.full((cap * world,), -9, dtype=torch.int32, device="cuda") for _ in range(nsteps)]
        torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
        shares = []
        for s, (descs, dev) in enumerate(steps):
            for o in range(world):
                if o != r:
                    mc.virtual_put(o, dev[o].data_ptr(), cap, stream=st_x.cuda_stream)
            mc.gather_enqueue_dev(dev[r].data_ptr(), cap, mode=0, stream=st_x.cuda_stream)
            shares.append(mc.match_enqueue_dev(40, outs[s].data_ptr(), cap * world, stream=sweep_stream.cuda_stream))
        torch.cuda.synchronize()
        assert abs(sum(nq for (_, _, _, nq, _) in shares[0]) - 28 * cap // world) <= 256           # 3.5 pairs' worth of query rows per rank, to one block
        for s in range(nsteps):
            res = outs[s].cpu().numpy()
            for (a, b, q0, nq, off) in shares[s]:
                got[s].setdefault((a, b), np.full(cap, -7, np.int32))[q0:q0 + nq] = res[off:off + nq]
        mc.close()
    for s in range(nsteps):
        assert sorted(got[s]) == multicam.exhaustive_pairs(world)
        for p in got[s]:
            assert np.array_equal(got[s][p], want[s][p]), (s, p)
    ctx.close()


def test_overlapped_steps_equal_one_stream_steps_with_ragged_counts(oracle):
    """clc_mc_set_overlap with counts that change every step (read on the device on odd steps), four ranks, six steps: exchange (and the
    filing of the peers' blocks) on one stream, sweeps on another, no host synchronisation in between -- every step's shares must be the
    one-stream handle's, hence the oracle's.  And the handle refuses to switch once it has exchanged."""
    import torch
    from coloc_amd import Context, MultiCam, CLCError
    world, cap = 4, 2600
    counts_by_step = [[2000, 1500, 2500, 1800], [2500, 2500, 0, 2100], [1999, 2001, 2003, 1], [2600, 2600, 2600, 2600], [300, 2600, 17, 900], [2048, 2047, 2049, 1024]]
    ctx = Context(device=0, width=160, height=120, maxkp=cap, detector=False)
    st_x, st_s = torch.cuda.Stream(), torch.cuda.Stream()
    steps = []
    for s, counts in enumerate(counts_by_step):
        descs = [synth.random_descriptors(n, seed=7000 + 13 * s + c) for c, n in enumerate(counts)]
        for c in range(1, world):
            k = min(len(descs[0]), len(descs[c])) // 2
            descs[c][:k] = descs[0][:k]
            if k:
                descs[c][:k, (c + s) % 64] ^= 0x3C
        dev = [torch.from_numpy(np.ascontiguousarray(np.concatenate([d, np.full((cap - len(d), 64), 0xA5, np.uint8)]))).cuda() for d in descs]
        d_cnt = [torch.tensor([n], dtype=torch.int32, device="cuda") for n in counts]
        steps.append((counts, descs, dev, d_cnt))
    torch.cuda.synchronize()
    for r in range(world):
        results = {}
        for overlap in (False, True):
            mc = MultiCam(ctx, world=world, rank=r, maxkp=cap)
            if overlap:
                mc.set_overlap(True)
            outs = [torch.full((cap * world,), -9, dtype=torch.int32, device="cuda") for _ in steps]
            torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
            shares = []
            for s, (counts, descs, dev, d_cnt) in enumerate(steps):
                for o in range(world):
                    if o != r:
                        mc.virtual_put(o, dev[o].data_ptr(), counts[o], stream=st_x.cuda_stream)
                if s % 2 == 0:
                    mc.gather_enqueue_dev(dev[r].data_ptr(), counts[r], mode=s % 2, stream=st_x.cuda_stream)
                else:
                    mc.gather_enqueue_dev(dev[r].data_ptr(), 0, mode=s % 2, stream=st_x.cuda_stream, d_my_count=d_cnt[r].data_ptr())
                shares.append(mc.match_enqueue_dev(40, outs[s].data_ptr(), cap * world, stream=(st_s if overlap else st_x).cuda_stream))
            torch.cuda.synchronize()
            if overlap:
                with pytest.raises(CLCError):
                    mc.set_overlap(False)
            results[overlap] = ([o.cpu().numpy() for o in outs], shares)
            mc.close()
        assert results[True][1] == results[False][1]
        for s, (counts, descs, dev, d_cnt) in enumerate(steps):
            assert np.array_equal(results[True][0][s], results[False][0][s]), (r, s)
            for (a, b, q0, nq, off) in results[True][1][s]:
                want = oracle.k2nn(descs[a], descs[b], 40) if counts[a] and counts[b] else np.full(counts[a], -1, np.int32)
                valid = max(0, min(nq, counts[a] - q0))
                assert np.array_equal(results[True][0][s][off:off + valid], want[q0:q0 + valid]), (r, s, a, b)
    ctx.close()
