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
