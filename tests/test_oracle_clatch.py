"""CPU tests of the CLATCH / pyramid / feeder oracle against (i) the learned-pattern hash, (ii) a
lane-level emulation of the reference reduce-transpose, (iii) closed-form cases and (iv) the
reference's own KFAST.h / FeatureAngle.h compiled into oracle/_ref."""
import hashlib
import math
import struct

import numpy as np
import pytest

import lane_emulation
import synth


def test_pattern_hash_matches_reference_table(oracle):
    """sha256 of the LE-u16 (a,b,c,0) table reconstructed from latch_pattern.inc == CLATCH.h:170's."""
    p = oracle.latch_pattern().astype(np.int64)
    vals = []
    for r in p:
        vals += [r[0] * 72 + r[1], r[2] * 72 + r[3], r[4] * 72 + r[5], 0]
    h = hashlib.sha256(struct.pack("<2048H", *vals)).hexdigest()
    assert h == "bef13cc5906f958322154c04973627e964e424a905273684af245904a31e4fd0"
    assert p.min() >= 5 and p.max() <= 53          # patches stay inside rows/cols 5..60


def test_pyramid_dims_match_survey(oracle):
    w, h, f = oracle.pyramid_dims(640, 480)
    assert list(zip(w, h)) == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    w, h, _ = oracle.pyramid_dims(1280, 720)
    assert list(zip(w, h)) == [(1280, 720), (1067, 600), (889, 500), (741, 417), (617, 347), (514, 289), (429, 241), (357, 201)]
    assert synth.pyramid_dims(640, 480)[0] == [640, 533, 444, 370, 309, 257, 214, 179]


def test_lerp_constant_and_range(oracle):
    for v in (0, 1, 127, 254, 255):
        img = np.full((48, 64), v, np.uint8)
        out = oracle.lerp(img, 1.2, 53, 40)
        assert (out == v).all()
    img = synth.rect_image(64, 48, n_rect=20, seed=3)
    out = oracle.lerp(img, 1.44, 44, 33)
    assert out.min() >= img.min() and out.max() <= img.max()


def test_lerp_matches_numpy_restatement(oracle):
    img = synth.rect_image(160, 120, n_rect=60, seed=5, noise_sigma=3.0)
    f = np.float32(np.float32(1.2) * np.float32(1.2))
    neww, newh = 111, 83
    out = oracle.lerp(img, float(f), neww, newh)
    f32 = np.float32
    ys = (np.arange(newh, dtype=f32) + f32(0.5)) * f - f32(0.5)
    xs = (np.arange(neww, dtype=f32) + f32(0.5)) * f - f32(0.5)
    j = np.floor(ys).astype(int); i = np.floor(xs).astype(int)
    wy = (ys - np.floor(ys)).astype(f32)[:, None]; wx = (xs - np.floor(xs)).astype(f32)[None, :]
    T = img.astype(f32) / f32(255.0)
    cl = lambda a, hi: np.clip(a, 0, hi)
    t00 = T[cl(j, 119)[:, None], cl(i, 159)[None, :]]; t10 = T[cl(j, 119)[:, None], cl(i + 1, 159)[None, :]]
    t01 = T[cl(j + 1, 119)[:, None], cl(i, 159)[None, :]]; t11 = T[cl(j + 1, 119)[:, None], cl(i + 1, 159)[None, :]]
    xa = (f32(1) - wx) * t00 + wx * t10
    xb = (f32(1) - wx) * t01 + wx * t11
    res = f32(255.0) * ((f32(1) - wy) * xa + wy * xb) + f32(0.5)
    assert np.array_equal(out, res.astype(np.uint8))


def _kp(x, y, angle, scale=0):
    k = np.zeros(1, dtype=synth.KP_DTYPE)
    k["x"], k["y"], k["angle"], k["scale"] = x, y, angle, scale
    return k


def test_roi_identity_rotation_and_clamp(oracle):
    img = synth.rect_image(200, 150, n_rect=80, seed=9, noise_sigma=2.0)
    roi = oracle.clatch_roi(img, _kp(100, 75, 0.0))
    assert np.array_equal(roi, img[75 - 32:75 + 32, 100 - 32:100 + 32])       # angle 0: plain crop
    # border keypoint: clamp-to-edge (GPUDetector.hpp:96-97)
    roi = oracle.clatch_roi(img, _kp(3, 3, 0.0))
    yy = np.clip(np.arange(-29, 35), 0, 149); xx = np.clip(np.arange(-29, 35), 0, 199)
    assert np.array_equal(roi, img[yy[:, None], xx[None, :]])
    # pi rotation = point reflection up to the +0.5 truncation asymmetry: check against the formula
    a = np.float32(math.pi)
    s = np.float32(np.sin(np.float64(a))); c = np.float32(np.cos(np.float64(a)))
    roi = oracle.clatch_roi(img, _kp(100, 75, float(a)))
    f32 = np.float32
    r, cc = np.mgrid[0:64, 0:64]
    xo = (cc - 32).astype(f32); yo = (r - 32).astype(f32)
    sx = ((f32(100) + (xo * c - yo * s)) + f32(0.5)).astype(np.int32)
    sy = ((f32(75) + (xo * s + yo * c)) + f32(0.5)).astype(np.int32)
    assert np.array_equal(roi, img[np.clip(sy, 0, 149), np.clip(sx, 0, 199)])


def test_descriptor_bits_match_lane_emulation(oracle):
    """Order-free S_n < 0 restatement == lane-level emulation of CLATCH.cu:169-188 (pixel split,
    reduce-transpose, bit placement), on real sampled windows."""
    img = synth.rect_image(320, 240, n_rect=150, seed=11, noise_sigma=4.0)
    pyr = oracle.pyramid(img)
    kps = synth.random_keypoints(6, 320, 240, seed=13)
    desc = oracle.clatch(pyr, kps)
    p = oracle.latch_pattern().astype(np.int64)
    trip = np.stack([p[:, 0] * 72 + p[:, 1], p[:, 2] * 72 + p[:, 3], p[:, 4] * 72 + p[:, 5]], 1)
    for k in range(len(kps)):
        roi = oracle.clatch_roi(pyr[int(kps[k]["scale"])], kps[k:k + 1])
        roi72 = np.zeros((64, 72), np.uint8); roi72[:, :64] = roi
        words = lane_emulation.clatch_bits_emulated(roi72, trip)
        assert np.array_equal(words, desc[k].view("<u4"))


def test_descriptor_is_not_degenerate(oracle):
    img = synth.rect_image(640, 480, seed=1000, noise_sigma=2.0)
    pyr = oracle.pyramid(img)
    kps = synth.random_keypoints(200, 640, 480, seed=2000)
    d = oracle.clatch(pyr, kps)
    ones = np.unpackbits(d, axis=1).mean()
    assert 0.2 < ones < 0.8
    assert len({bytes(r) for r in d}) > 150


def test_fast9_and_angle_match_compiled_reference(oracle, ref_feeder):
    """orc_fast9 / orc_feature_angle == the reference's KFAST<mt,true> / featureAngle compiled from
    /root/reference (oracle/_ref), including the width = 38 + 16k shift-by-32 quirk."""
    for (W, H, seed) in [(640, 480, 1000), (214, 161, 5), (179, 134, 6), (257, 193, 7), (215, 60, 8)]:
        img = synth.rect_image(W, H, seed=seed, noise_sigma=2.0)
        a = oracle.fast9(img, 40)
        for mt in (True, False):
            b = ref_feeder.kfast(img, 40, mt)
            assert len(a) == len(b) > 0
            for f in ("x", "y", "score"):
                assert np.array_equal(a[f], b[f])
        ao = np.array([oracle.feature_angle(img, int(k["x"]), int(k["y"])) for k in a[:200]])
        ar = np.array([ref_feeder.feature_angle(img, int(k["x"]), int(k["y"])) for k in a[:200]])
        assert np.array_equal(ao.view(np.uint32), ar.view(np.uint32))


def test_fast9_width_quirk(oracle, ref_feeder):
    rng = np.random.default_rng(1)
    for W, expect_any in ((214, False), (215, True)):
        base = np.zeros((60, W), np.int32); base[10:21, 185:196] = 200
        img = (base + rng.integers(0, 30, size=base.shape) * (base > 0)).astype(np.uint8)
        a = oracle.fast9(img, 40); b = ref_feeder.kfast(img, 40, False)
        assert (len(a) > 0) == expect_any and len(a) == len(b)


def edge_image(W, H):
    """the noisy checkerboard of tests/test_gpu_detect.py::test_small_and_edge_sizes"""
    rng = np.random.default_rng(W * 1000 + H)
    yy, xx = np.mgrid[0:H, 0:W]
    img = (((yy // 5 + xx // 7) % 2) * 150 + 40 + rng.integers(0, 25, size=(H, W))).astype(np.uint8)
    img[rng.random((H, W)) < 0.02] = 255
    return img


def test_fast9_edge_sizes_match_compiled_reference(oracle, ref_feeder):
    """The sizes the GPU detector is tested at around its tile seams and at widths 6 (mod 16) (38 = the narrowest image whose row walk
    lands on cols - 35): the restatement == the compiled reference there too, so the GPU test's oracle is pinned at those sizes."""
    dropped = 0
    for (W, H) in [(38, 40), (22, 30), (54, 33), (70, 17), (134, 50), (64, 16), (65, 17), (129, 48), (198, 64), (1286, 40)]:
        img = edge_image(W, H)
        a = oracle.fast9(img, 40)
        b = ref_feeder.kfast(img, 40, False)
        assert len(a) == len(b)
        for f in ("x", "y", "score"):
            assert np.array_equal(a[f], b[f])
        if W % 16 == 6 and W >= 38:
            dropped += int((a["x"] >= W - 35).sum() == 0 or W == 38)
    assert dropped >= 1


def test_features_conversion(oracle):
    kps = synth.random_keypoints(50, 640, 480, seed=3)
    f = oracle.features_from_kps(kps)
    sc = np.array([np.float32(np.float64(np.float32(1.2)) ** int(s)) for s in kps["scale"]], dtype=np.float32)
    assert np.array_equal(f[:, 0], sc * kps["x"].astype(np.float32))
    assert np.array_equal(f[:, 2], np.float32(7.0) * sc)
    assert np.array_equal(f[:, 3], kps["angle"])
