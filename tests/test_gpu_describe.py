"""GPU parity of the pyramid (LERP) and CLATCH kernels through the C ABI: bytes / descriptor bits
identical to the CPU oracle and to the golden fixtures.  Reference: src/CUDALERP.cu:157-178,
src/CLATCH.cu:157-188, include/coloc/GPUDetector.hpp:109-114,232-255."""
import math
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ctx(w, h, maxkp=20000):
    from coloc_amd import Context
    return Context(device=0, width=w, height=h, maxkp=maxkp)


@pytest.mark.parametrize("W,H", [(640, 480), (1280, 720), (160, 120), (101, 67)])
@pytest.mark.parametrize("kind", ["rect", "noise", "gradient", "const"])
def test_pyramid_bytes_identical(oracle, W, H, kind):
    if kind == "rect":
        img = synth.rect_image(W, H, seed=1000, noise_sigma=2.0)
    elif kind == "noise":
        img = np.random.default_rng(5).integers(0, 256, (H, W), dtype=np.uint8)
    elif kind == "gradient":
        img = ((np.arange(W)[None, :] * 255 // (W - 1)) * np.ones((H, 1))).astype(np.uint8)
    else:
        img = np.full((H, W), 255, np.uint8)
    ctx = _ctx(W, H, 1024)
    ctx.pyramid_build(img)
    pyr = oracle.pyramid(img)
    ws, hs, _ = oracle.pyramid_dims(W, H)
    for lv in range(8):
        w, h, pitch, _ = ctx.pyramid_level(lv)
        assert (w, h) == (ws[lv], hs[lv]) and pitch % 64 == 0
        assert np.array_equal(ctx.pyramid_download(lv), pyr[lv]), "level %d" % lv
    ctx.close()


def test_clatch_golden_fixture(oracle):
    g = np.load(os.path.join(G, "clatch_160x120.npz"))
    ctx = _ctx(160, 120, 1024)
    ctx.pyramid_build(g["img"])
    for i in range(8):
        assert np.array_equal(ctx.pyramid_download(i), g["level%d" % i])
    kps = g["kps"].reshape(-1).view(synth.KP_DTYPE)
    assert np.array_equal(ctx.describe(kps), g["desc"])
    ctx.close()


@pytest.mark.parametrize("W,H,n,seed", [(640, 480, 10000, 2000), (640, 480, 10000, 2001), (1280, 720, 6000, 2002), (160, 120, 777, 2003)])
def test_clatch_bits_identical(oracle, W, H, n, seed):
    img = synth.rect_image(W, H, seed=1000 + seed, noise_sigma=2.0)
    ctx = _ctx(W, H, 20000)
    ctx.pyramid_build(img)
    pyr = oracle.pyramid(img)
    kps = synth.random_keypoints(n, W, H, seed=seed)
    d = ctx.describe(kps)
    do = oracle.clatch(pyr, kps)
    bad = np.nonzero((d != do).any(1))[0]
    assert bad.size == 0, "descriptors differ for keypoints %s" % bad[:10]
    ctx.close()


def test_clatch_borders_all_scales_special_angles(oracle):
    W, H = 640, 480
    img = synth.rect_image(W, H, seed=77, noise_sigma=3.0)
    ctx = _ctx(W, H, 4096)
    ctx.pyramid_build(img)
    pyr = oracle.pyramid(img)
    ws, hs, _ = oracle.pyramid_dims(W, H)
    rows = []
    angles = [0.0, math.pi / 2, -math.pi / 2, math.pi, -math.pi, math.pi / 4, 3 * math.pi / 4, 1e-7, -1e-7, 2.5, -0.7]
    for lv in range(8):
        for (x, y) in [(0, 0), (3, 3), (ws[lv] - 1, hs[lv] - 1), (ws[lv] - 4, 3), (3, hs[lv] - 4), (ws[lv] // 2, hs[lv] // 2)]:
            for a in angles:
                rows.append((x, y, 0, np.float32(a), lv))
    kps = np.array(rows, dtype=[("x", "<i4"), ("y", "<i4"), ("score", "u1"), ("angle", "<f4"), ("scale", "u1")])
    k2 = np.zeros(len(kps), dtype=synth.KP_DTYPE)
    for f in ("x", "y", "score", "angle", "scale"):
        k2[f] = kps[f]
    assert np.array_equal(ctx.describe(k2), oracle.clatch(pyr, k2))
    ctx.close()


def test_describe_errors(gpu_ctx):
    from coloc_amd import Context, CLCError
    ctx = Context(device=0, width=160, height=120, maxkp=16)
    kps = synth.random_keypoints(8, 160, 120, seed=1)
    with pytest.raises(CLCError) as e:          # describe before pyramid_build
        ctx.describe(kps)
    assert e.value.status == 5
    ctx.pyramid_build(synth.rect_image(160, 120, n_rect=30, seed=1))
    with pytest.raises(CLCError) as e:          # capacity
        ctx.describe(synth.random_keypoints(17, 160, 120, seed=1))
    assert e.value.status == 2
    with pytest.raises(CLCError) as e:          # wrong image size
        ctx.pyramid_build(np.zeros((100, 100), np.uint8))
    assert e.value.status == 1
    assert ctx.describe(kps[:0]).shape == (0, 64)
    ctx.close()


def test_full_size_describe_then_match_roundtrip(oracle):
    """config[1] end to end on the GPU vs the oracle: 2 images x 10k keypoints -> descriptors -> K2NN."""
    W, H, n = 640, 480, 10000
    ctx = _ctx(W, H, n)
    descs, odescs = [], []
    for i in range(2):
        img = synth.rect_image(W, H, seed=1000 + i, noise_sigma=2.0)
        ctx.pyramid_build(img)
        kps = synth.random_keypoints(n, W, H, seed=2000 + i)
        descs.append(ctx.describe(kps))
        odescs.append(oracle.clatch(oracle.pyramid(img), kps))
    assert np.array_equal(descs[0], odescs[0]) and np.array_equal(descs[1], odescs[1])
    assert np.array_equal(ctx.match_2nn(descs[0], descs[1], 40), oracle.k2nn(odescs[0], odescs[1], 40))
    ctx.close()


def test_pyramid_from_device_image_with_pitch(oracle):
    """clc_pyramid_build_dev: level 0 comes from a caller-owned device image with pitch > width; the
    copy into the arena and the 7 resamples are one launch."""
    import torch
    W, H = 640, 480
    img = synth.rect_image(W, H, seed=5, noise_sigma=2.0)
    pitch = 704
    padded = np.full((H, pitch), 0xAB, np.uint8)
    padded[:, :W] = img
    d = torch.from_numpy(padded).cuda()
    ctx = _ctx(W, H, 64)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ctx.pyramid_build_dev(d.data_ptr(), W, H, pitch, st.cuda_stream)
    st.synchronize()
    pyr = oracle.pyramid(img)
    for lv in range(8):
        assert np.array_equal(ctx.pyramid_download(lv), pyr[lv]), "level %d" % lv
    ctx.close()


def test_more_than_65536_keypoints_grid_stride(oracle):
    """n > 65536: the one-wave-per-keypoint grid wraps and waves stride over the keypoints."""
    W, H, n = 640, 480, 70001
    img = synth.rect_image(W, H, seed=31, noise_sigma=2.0)
    kps = synth.random_keypoints(n, W, H, seed=32)
    ctx = _ctx(W, H, n)
    ctx.pyramid_build(img)
    d = ctx.describe(kps)
    pyr = oracle.pyramid(img)
    idx = np.concatenate([np.arange(0, 2000), np.arange(65000, 67000), np.arange(n - 2000, n)])
    assert np.array_equal(d[idx], oracle.clatch(pyr, kps[idx]))
    assert len({bytes(r) for r in d[::97]}) > 500
    ctx.close()


def test_batched_cameras_equal_per_image_calls(oracle):
    """clc_describe_batch_dev (one pyramid + one CLATCH launch for several cameras) == the per-image path, bit for
    bit, including a camera without keypoints, ragged counts and a pitched source image; camera 0's pyramid is the
    context's current one afterwards, and growing the batch on the same context works."""
    import torch
    W, H = 320, 240
    pitch = 384
    ctx, single = _ctx(W, H, 4096), _ctx(W, H, 4096)
    for n_cam, counts in [(3, [1500, 0, 2500]), (5, [7, 64, 1, 3000, 129])]:
        imgs = [synth.rect_image(W, H, seed=4000 + 10 * n_cam + c, noise_sigma=2.0) for c in range(n_cam)]
        kps = [synth.random_keypoints(counts[c], W, H, seed=4100 + 10 * n_cam + c) for c in range(n_cam)]
        d_imgs = []
        for im in imgs:
            t = torch.zeros((H, pitch), dtype=torch.uint8, device="cuda:0")
            t[:, :W] = torch.from_numpy(im).cuda()
            d_imgs.append(t)
        d_kps = [torch.from_numpy(k.view(np.uint8).reshape(-1, 20).copy()).cuda() if len(k) else torch.zeros((1, 20), dtype=torch.uint8, device="cuda:0") for k in kps]
        d_desc = [torch.full((max(c, 1), 64), 0xAB, dtype=torch.uint8, device="cuda:0") for c in counts]
        torch.cuda.synchronize()      # the fill runs on torch's stream, the library on its own (non-blocking) one: order them
        torch.cuda.synchronize()
        ctx.describe_batch_dev([t.data_ptr() for t in d_imgs], W, H, pitch, [t.data_ptr() for t in d_kps], counts,
                               [t.data_ptr() for t in d_desc])
        ctx.sync()
        for c in range(n_cam):
            single.pyramid_build(imgs[c])
            want = single.describe(kps[c]) if counts[c] else np.zeros((0, 64), np.uint8)
            got = d_desc[c].cpu().numpy()[:counts[c]]
            assert np.array_equal(got, want), "camera %d of %d" % (c, n_cam)
            if counts[c] == 0:
                assert (d_desc[c].cpu().numpy() == 0xAB).all()          # nothing written for an empty camera
        pyr0 = oracle.pyramid(imgs[0])
        for lv in range(8):
            assert np.array_equal(ctx.pyramid_download(lv), pyr0[lv])
        # and the oracle itself on one camera of the batch
        c = n_cam - 2
        assert np.array_equal(d_desc[c].cpu().numpy()[:counts[c]], oracle.clatch(oracle.pyramid(imgs[c]), kps[c]))
    with pytest.raises(Exception):
        ctx.describe_batch_dev([0] * 9, W, H, pitch, [0] * 9, [0] * 9, [0] * 9)
    ctx.close(); single.close()
