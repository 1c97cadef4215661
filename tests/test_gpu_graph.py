"""The `_dev` entry points only enqueue on the caller's stream (no allocation, no synchronisation, job lists as kernel arguments), so a
step built from them can be captured into a hipGraph and replayed with other inputs in the same buffers (include/coloc_hip.h,
DESIGN.md section 1).  Captured here through torch.cuda.graph on the stream the calls are given; every replay is checked against the oracle."""
import numpy as np
import pytest

import synth
from test_gpu_detect import oracle_detect, same_kps

pytestmark = pytest.mark.gpu


def test_describe_match_step_captured_and_replayed(oracle):
    """clc_describe_batch_dev + clc_match_jobs_dev + clc_match_map_dev in one graph; three replays, each with new images and keypoints
    written into the captured buffers: descriptors, pair matches and map matches == the oracle's every time."""
    import torch
    from coloc_amd import Context, multicam
    W, H, N, M = 320, 240, 1500, 1200
    ctx = Context(device=0, width=W, height=H, maxkp=N, match_thresh=60)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    sp = st.cuda_stream
    imgs = [torch.zeros((H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
    kps = [torch.zeros((N, 20), dtype=torch.uint8, device=dev) for _ in range(2)]
    arena = torch.zeros((2, N, 64), dtype=torch.uint8, device=dev)
    d_pair = torch.full((N,), -9, dtype=torch.int32, device=dev)
    d_map = torch.full((N,), -9, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
    jobs = multicam.jobs_to_abi(multicam.shard_pairs([N, N], 1, 0, grain=ctx.k2nn_queries_per_block), [N, N], N, 40)

    def fill(rep):
        data = []
        for c in range(2):
            img = synth.rect_image(W, H, n_rect=150, seed=9000 + 10 * rep + c, noise_sigma=2.0)
            kp = synth.random_keypoints(N, W, H, seed=9100 + rep)            # the same keypoints in both frames: real matches
            imgs[c].copy_(torch.from_numpy(img)); kps[c].copy_(torch.from_numpy(kp.view(np.uint8).reshape(-1, 20).copy()))
            data.append((img, kp))
        return data

    def step():
        ctx.describe_batch_dev([t.data_ptr() for t in imgs], W, H, W, [t.data_ptr() for t in kps], [N, N], [arena[0].data_ptr(), arena[1].data_ptr()], sp)
        ctx.match_jobs_dev(arena.data_ptr(), jobs, d_pair.data_ptr(), sp)
        ctx.match_map_dev(arena[1].data_ptr(), N, 60, d_map.data_ptr(), sp)

    fill(0)
    map_desc = synth.random_descriptors(M, seed=77)
    ctx.set_map(map_desc)
    with torch.cuda.stream(st):
        step()                                                               # eager once: the arena grows to two pyramids here, not in the capture
    torch.cuda.synchronize()
    first = arena[0, :M].cpu().numpy().copy()
    map_desc[:M // 2] = first[:M // 2]                                       # half of the map = descriptors of replay 0's first frame
    ctx.set_map(map_desc)                                                    # same size, same device buffer: the captured sweep reads the new rows
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        step()
    for rep in range(3):
        data = fill(rep)
        arena.zero_(); d_pair.fill_(-9); d_map.fill_(-9)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        want = [oracle.clatch(oracle.pyramid(img), kp) for img, kp in data]
        got = arena.cpu().numpy()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), rep
        assert np.array_equal(d_pair.cpu().numpy(), oracle.k2nn(want[0], want[1], 40)), rep
        wm = oracle.k2nn(want[1], map_desc, 60)
        assert np.array_equal(d_map.cpu().numpy(), wm), rep
        if rep == 0:
            assert (oracle.k2nn(want[0], map_desc, 60) >= 0).sum() >= M // 2 - 5
    ctx.close()


def test_detect_describe_counted_match_captured_and_replayed(oracle):
    """The real front end in a graph: clc_detect_batch_dev (pyramid, two detector launches, CLATCH with the counts read on the device)
    + clc_match_jobs_counted_dev (the pair sweep with the counts read on the device).  Three replays with new frames: keypoints,
    counts, descriptors and matches == oracle(detect -> describe -> match)."""
    import torch
    from coloc_amd import Context
    from coloc_amd.abi import KP_DTYPE
    W, H, CAP = 320, 240, 6000
    ctx = Context(device=0, width=W, height=H, maxkp=CAP)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    sp = st.cuda_stream
    imgs = [torch.zeros((H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
    kps = [torch.zeros((CAP, 20), dtype=torch.uint8, device=dev) for _ in range(2)]
    cnt = [torch.zeros((2,), dtype=torch.int32, device=dev) for _ in range(2)]
    arena = torch.zeros((2, CAP, 64), dtype=torch.uint8, device=dev)
    d_pair = torch.full((CAP,), -9, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them

    def step():
        ctx.detect_batch_dev([t.data_ptr() for t in imgs], W, H, W, [t.data_ptr() for t in kps], [t.data_ptr() for t in cnt],
                             [arena[0].data_ptr(), arena[1].data_ptr()], sp)
        ctx.match_jobs_counted_dev(arena.data_ptr(), [(0, CAP, CAP, CAP, 0, 40)], [cnt[0].data_ptr()], [cnt[1].data_ptr()], [0], d_pair.data_ptr(), sp)

    def fill(rep):
        base = synth.rect_image(W, H, n_rect=100 + 40 * rep, seed=9300 + rep, noise_sigma=0.0).astype(np.float32)
        out = []
        for c in range(2):
            img = np.clip(base + np.random.default_rng(9400 + 2 * rep + c).normal(0.0, 2.0, base.shape) + 0.5, 0, 255).astype(np.uint8)
            imgs[c].copy_(torch.from_numpy(img))
            out.append(img)
        return out

    fill(0)
    with torch.cuda.stream(st):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        step()
    for rep in range(3):
        frames = fill(rep)
        d_pair.fill_(-9)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        descs = []
        for c in range(2):
            pyr, want = oracle_detect(oracle, frames[c])
            n = int(cnt[c][0].item())
            assert n == len(want) < CAP and int(cnt[c][1].item()) == len(want)
            assert same_kps(kps[c].cpu().numpy().reshape(-1).view(KP_DTYPE)[:n], want)
            d = oracle.clatch(pyr, want)
            assert np.array_equal(arena[c, :n].cpu().numpy(), d)
            descs.append(d)
        res = d_pair.cpu().numpy()
        assert np.array_equal(res[:len(descs[0])], oracle.k2nn(descs[0], descs[1], 40))
        assert (res[len(descs[0]):] == -1).all()                            # planned rows past the camera's count
        assert (res[:len(descs[0])] >= 0).sum() > 100                        # the two frames see the same scene
    ctx.close()
