"""GPU detector (FAST-9 + NMS + orientation, coloc_amd/csrc/detect.hip) against the oracle's
restatement of KFAST.h / FeatureAngle.h -- which itself is pinned to the compiled reference
(tests/test_oracle_clatch.py, tests/golden/feeder_ref.npz) -- and the whole GPU-resident front end
(detect -> describe) against oracle(detect) -> oracle(describe).  Keypoint order, coordinates,
scores and angle bits must be identical (GPUDetector.hpp:262-277)."""
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def oracle_detect(oracle, img, thresh=40, levels=8):
    pyr = oracle.pyramid(img, levels=levels)
    out = []
    for lv, im in enumerate(pyr):
        k = oracle.fast9(im, thresh)
        k["scale"] = lv
        for i in range(len(k)):
            k["angle"][i] = oracle.feature_angle(im, int(k["x"][i]), int(k["y"][i]))
        out.append(k)
    return pyr, np.concatenate(out)


def same_kps(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in ("x", "y", "score", "scale")) and \
        np.array_equal(a["angle"].view(np.uint32), b["angle"].view(np.uint32))


@pytest.mark.parametrize("W,H,seed", [(640, 480, 1000), (640, 480, 1001), (1280, 720, 1002), (320, 240, 6), (214, 161, 5), (230, 100, 9)])
def test_detect_matches_oracle(oracle, W, H, seed):
    from coloc_amd import Context
    img = synth.rect_image(W, H, seed=seed, noise_sigma=2.0)
    ctx = Context(device=0, width=W, height=H, maxkp=60000)
    ctx.pyramid_build(img)
    kps, found = ctx.detect()
    _, want = oracle_detect(oracle, img)
    assert found == len(want) and len(want) > 100
    assert same_kps(kps, want)
    ctx.close()


def test_detect_reference_fixture(oracle):
    """Single-level check against GENUINE reference outputs (tests/golden/feeder_ref.npz), including the
    214-px-wide image that triggers KFAST.h:245's shift-by-32 quirk."""
    from coloc_amd import Context
    g = np.load(os.path.join(G, "feeder_ref.npz"))
    for name in ("a", "b"):
        img = g["img_" + name]
        H, W = img.shape
        ctx = Context(device=0, width=W, height=H, maxkp=20000, scale_levels=1)
        ctx.pyramid_build(img)
        kps, found = ctx.detect()
        assert np.array_equal(np.stack([kps["x"], kps["y"], kps["score"].astype(np.int32)], 1), g["xys_" + name])
        assert np.array_equal(kps["angle"].view(np.uint32), g["angle_" + name].view(np.uint32))
        ctx.close()


def test_width_quirk_rows(oracle):
    """cols = 214 (6 mod 16): rows whose 32-column walk lands on cols-35 lose their last 32 columns."""
    from coloc_amd import Context
    rng = np.random.default_rng(1)
    for W in (214, 215):
        base = np.zeros((60, W), np.int32); base[10:21, 185:196] = 200; base[30:41, 26:32] = 180; base[30:41, 190:200] = 210
        img = (base + rng.integers(0, 30, size=base.shape) * (base > 0)).astype(np.uint8)
        ctx = Context(device=0, width=W, height=60, maxkp=4096, scale_levels=1)
        ctx.pyramid_build(img)
        kps, _ = ctx.detect()
        want = oracle.fast9(img, 40)
        assert np.array_equal(kps["x"], want["x"]) and np.array_equal(kps["y"], want["y"]) and np.array_equal(kps["score"], want["score"])
        ctx.close()


def test_capacity_truncates_in_order(oracle):
    from coloc_amd import Context
    img = synth.rect_image(640, 480, seed=1000, noise_sigma=2.0)
    _, want = oracle_detect(oracle, img)
    ctx = Context(device=0, width=640, height=480, maxkp=500)
    ctx.pyramid_build(img)
    kps, found = ctx.detect()
    assert found == len(want) and len(kps) == 500 and same_kps(kps, want[:500])
    ctx.close()


def test_detect_and_describe_front_end(oracle):
    """GPU-resident detect -> describe == oracle detect -> oracle describe, descriptors bit for bit."""
    from coloc_amd import Context
    W, H = 640, 480
    img = synth.rect_image(W, H, seed=1003, noise_sigma=2.0)
    ctx = Context(device=0, width=W, height=H, maxkp=20000)
    kps, desc, found = ctx.detect_and_describe(img)
    pyr, want = oracle_detect(oracle, img)
    assert same_kps(kps, want) and found == len(want)
    assert np.array_equal(desc, oracle.clatch(pyr, want))
    ctx.close()


@pytest.mark.parametrize("W,H", [(38, 40), (22, 30), (54, 33), (70, 17), (134, 50), (64, 16), (65, 17), (129, 48), (198, 64), (1286, 40)])
def test_small_and_edge_sizes(oracle, W, H):
    """Single level at widths around the tile size (64) and at widths 6 (mod 16) -- 38 is the narrowest image whose walk can land on
    cols - 35 (every row does, at once), 1286 makes the replay cross many 64-column chunks -- against the oracle (which is pinned to the
    compiled reference): a noisy checkerboard so that corners sit everywhere, borders and tile seams included."""
    from coloc_amd import Context
    from test_oracle_clatch import edge_image
    img = edge_image(W, H)
    ctx = Context(device=0, width=W, height=H, maxkp=30000, scale_levels=1)
    ctx.pyramid_build(img)
    kps, found = ctx.detect()
    want = oracle.fast9(img, 40)
    assert found == len(want)
    assert np.array_equal(kps["x"], want["x"]) and np.array_equal(kps["y"], want["y"]) and np.array_equal(kps["score"], want["score"])
    ctx.close()


def test_dense_corners_every_tile(oracle):
    """A frame where a large share of the pixels pass the pre-test (salt-and-pepper noise): the candidate lists of the tiles run long
    (several passes of the ring stage), all 8 levels."""
    from coloc_amd import Context
    W, H = 640, 480
    rng = np.random.default_rng(77)
    img = rng.integers(90, 110, size=(H, W)).astype(np.uint8)
    m = rng.random((H, W))
    img[m < 0.04] = 255
    img[m > 0.96] = 0
    ctx = Context(device=0, width=W, height=H, maxkp=200000)
    ctx.pyramid_build(img)
    kps, found = ctx.detect()
    _, want = oracle_detect(oracle, img)
    assert found == len(want) and len(want) > 5000
    assert same_kps(kps, want)
    ctx.close()


@pytest.mark.parametrize("W,H,n", [(640, 480, 4), (214, 161, 8), (1280, 720, 2)])
def test_detect_batch_equals_single_camera_calls(oracle, W, H, n):
    """clc_detect_batch_dev (one pyramid launch, two detector launches, one CLATCH launch for n cameras) == the oracle's
    detect -> describe of every frame: keypoints, order, counts, descriptors.  A second call with the frames in another order reuses
    the buffers (stale masks / counts of the call before must not leak)."""
    import torch
    from coloc_amd import Context
    from coloc_amd.abi import KP_DTYPE
    cap = 12000
    ctx = Context(device=0, width=W, height=H, maxkp=cap)
    imgs = [synth.rect_image(W, H, seed=1200 + c, noise_sigma=2.0 + c) for c in range(n)]
    want = [oracle_detect(oracle, im) for im in imgs]
    d_kps = [torch.zeros((cap, 20), dtype=torch.uint8, device="cuda") for _ in range(n)]
    d_cnt = [torch.zeros((2,), dtype=torch.int32, device="cuda") for _ in range(n)]
    d_desc = [torch.zeros((cap, 64), dtype=torch.uint8, device="cuda") for _ in range(n)]
    for order in (list(range(n)), list(reversed(range(n)))):
        d_imgs = [torch.from_numpy(imgs[c]).cuda() for c in order]
        ctx.detect_batch_dev([t.data_ptr() for t in d_imgs], W, H, W, [t.data_ptr() for t in d_kps], [t.data_ptr() for t in d_cnt],
                             [t.data_ptr() for t in d_desc])
        ctx.sync()
        for b, c in enumerate(order):
            pyr, w = want[c]
            cnt = d_cnt[b].cpu().numpy()
            assert cnt[0] == min(len(w), cap) and cnt[1] == len(w)
            kps = d_kps[b].cpu().numpy().reshape(-1).view(KP_DTYPE)[:cnt[0]]
            assert same_kps(kps, w[:cnt[0]])
            assert np.array_equal(d_desc[b].cpu().numpy()[:cnt[0]], oracle.clatch(pyr, w[:cnt[0]]))
    ctx.close()


def test_width_beyond_the_detector_limit_is_refused_at_creation():
    """CLC_DETECT_MAX_WIDTH (4096) was only enforced inside the detector launch, so a wider context was created and
    every later clc_detect* call failed with a bare hipErrorInvalidValue.  clc_ctx_create refuses it with CLC_ERR_CAPACITY (2);
    the widest allowed frame still detects like the oracle."""
    from coloc_amd import Context, CLCError
    with pytest.raises(CLCError) as e:
        Context(device=0, width=4097, height=64, maxkp=1000)
    assert e.value.status == 2
    ctx = Context(device=0, width=4096, height=40, maxkp=4000)
    ctx.close()
