"""The pair step (clc_describe_match_pair_dev): describe both cameras of a pair and match them as one enqueue.  It replaces
detectAndDescribe x 2 (GPUDetector.hpp:216-291) + computeMatchesPair (GPUMatcher.hpp:165-172): descriptors must equal the oracle's CLATCH
bit for bit, the match indices the oracle's K2NN on those descriptors -- for ragged counts, empty sides, repeated steps (the top-2 rows
and arrival counters re-arm themselves) and both formulations of the sweep.  (Round 5's chunked form with its device-side gates is gone
from the library: profiles/r06_removed_variants.patch.)"""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
W, H = 640, 480


def _inputs(torch, nq, nt, seed):
    scene = synth.rect_image(W, H, seed=1000 + seed, noise_sigma=0.0).astype(np.float32)
    imgs = [np.clip(scene + np.random.default_rng(seed * 7 + c).normal(0.0, 2.0, scene.shape) + 0.5, 0, 255).astype(np.uint8) for c in range(2)]
    base = synth.random_keypoints(max(nq, nt, 1), W, H, seed=2000 + seed)
    kps = [base[np.random.default_rng(seed * 11 + c).permutation(len(base))][:n] for c, n in enumerate((nq, nt))]
    d_imgs = [torch.from_numpy(i).cuda() for i in imgs]
    d_kps = [torch.from_numpy(k.view(np.uint8).reshape(-1, 20).copy()).cuda() if len(k) else torch.zeros((1, 20), dtype=torch.uint8, device="cuda") for k in kps]
    return imgs, kps, d_imgs, d_kps


@pytest.mark.parametrize("formulation", ["matrix", "popcount"])
@pytest.mark.parametrize("nq,nt", [(10000, 10000), (9000, 10000), (4097, 3000), (777, 5000), (2048, 2048), (6200, 1), (0, 100), (100, 0)])
def test_pair_step_equals_oracle(oracle, formulation, nq, nt):
    import torch
    from coloc_amd import Context
    ctx = Context(device=0, width=W, height=H, maxkp=10000)
    ctx.set_k2nn_formulation(formulation)
    try:
        imgs, kps, d_imgs, d_kps = _inputs(torch, nq, nt, seed=nq % 97 + nt % 89)
        pyr = [oracle.pyramid(i) for i in imgs]
        want_d = [oracle.clatch(pyr[c], kps[c]) if len(kps[c]) else np.zeros((0, 64), np.uint8) for c in range(2)]
        want_m = oracle.k2nn(want_d[0], want_d[1], 40) if nq and nt else np.full(nq, -1, np.int32)
        assert nq < 3000 or nt < 3000 or (want_m >= 0).sum() > nq // 4              # the cameras see the same scene: the accept branch is exercised
        desc = [torch.full((max(n, 1), 64), 0xA5, dtype=torch.uint8, device="cuda") for n in (nq, nt)]
        match = torch.full((max(nq, 1),), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
        torch.cuda.synchronize()
        for rep in range(3):                                                        # rows and arrival words re-arm
            ctx.describe_match_pair_dev([t.data_ptr() for t in d_imgs], W, H, W, [t.data_ptr() for t in d_kps], [nq, nt],
                                        [t.data_ptr() for t in desc], 40, match.data_ptr())
        ctx.sync()
        for c, n in enumerate((nq, nt)):
            assert np.array_equal(desc[c].cpu().numpy()[:n], want_d[c]), c
        assert np.array_equal(match.cpu().numpy()[:nq], want_m)
    finally:
        ctx.close()


def test_pair_step_back_to_back_with_changing_inputs(oracle):
    """40 steps enqueued without a host synchronisation in between, the keypoint order and the counts changing from step to step (each
    step writes its own output buffers): stream order alone keeps the steps apart."""
    import torch
    from coloc_amd import Context
    ctx = Context(device=0, width=W, height=H, maxkp=10000)
    try:
        imgs, kps, d_imgs, d_kps = _inputs(torch, 10000, 10000, seed=5)
        pyr = [oracle.pyramid(i) for i in imgs]
        full = [oracle.clatch(pyr[c], kps[c]) for c in range(2)]
        shapes = [(10000, 10000), (8000, 9000), (2500, 10000), (10000, 300), (5121, 5119)]
        outs = []
        for it in range(40):
            nq, nt = shapes[it % len(shapes)]
            desc = [torch.zeros((10000, 64), dtype=torch.uint8, device="cuda") for _ in range(2)]
            match = torch.full((10000,), -7, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
            ctx.describe_match_pair_dev([t.data_ptr() for t in d_imgs], W, H, W, [t.data_ptr() for t in d_kps], [nq, nt],
                                        [t.data_ptr() for t in desc], 40, match.data_ptr())
            outs.append((nq, nt, desc, match))
        ctx.sync()
        cache = {}
        for nq, nt, desc, match in outs:
            if (nq, nt) not in cache:
                cache[(nq, nt)] = oracle.k2nn(full[0][:nq], full[1][:nt], 40)
            assert np.array_equal(desc[0].cpu().numpy()[:nq], full[0][:nq]) and np.array_equal(desc[1].cpu().numpy()[:nt], full[1][:nt])
            assert np.array_equal(match.cpu().numpy()[:nq], cache[(nq, nt)])
    finally:
        ctx.close()


def test_pair_step_is_capturable(oracle):
    """Enqueue only, no allocation after the first call: the step is captured into a hipGraph and replayed with new images in the same
    buffers."""
    import torch
    from coloc_amd import Context
    ctx = Context(device=0, width=W, height=H, maxkp=6000)
    try:
        n = 6000
        imgs, kps, d_imgs, d_kps = _inputs(torch, n, n, seed=9)
        desc = [torch.zeros((n, 64), dtype=torch.uint8, device="cuda") for _ in range(2)]
        match = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
        st = torch.cuda.Stream()
        args = ([t.data_ptr() for t in d_imgs], W, H, W, [t.data_ptr() for t in d_kps], [n, n], [t.data_ptr() for t in desc], 40, match.data_ptr())
        with torch.cuda.stream(st):
            ctx.describe_match_pair_dev(*args, stream=st.cuda_stream)          # warm: workspace
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                ctx.describe_match_pair_dev(*args, stream=st.cuda_stream)
            for rep in range(3):
                img2 = [np.roll(i, 3 * rep + 1, axis=1).copy() for i in imgs]
                for t, i in zip(d_imgs, img2):
                    t.copy_(torch.from_numpy(i))
                match.fill_(-7)
                g.replay()
                torch.cuda.synchronize()
                pyr = [oracle.pyramid(i) for i in img2]
                want_d = [oracle.clatch(pyr[c], kps[c]) for c in range(2)]
                assert np.array_equal(desc[0].cpu().numpy(), want_d[0]) and np.array_equal(desc[1].cpu().numpy(), want_d[1])
                assert np.array_equal(match.cpu().numpy(), oracle.k2nn(want_d[0], want_d[1], 40))
    finally:
        ctx.close()
