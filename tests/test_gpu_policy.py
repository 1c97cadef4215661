"""HIPDetector<bool> / HIPMatcher<bool> driven like ColoC drives GPUDetector / GPUMatcher (C++ driver
tests/host/policy_driver.cpp) on the GPU, checked against the oracle: keypoints, features,
descriptors, and the three IndMatch index conventions (GPUMatcher.hpp:217,265)."""
import os
import subprocess

import numpy as np
import pytest

import synth
from test_policy_host import build_driver
from test_gpu_detect import oracle_detect, same_kps

pytestmark = pytest.mark.gpu


def _pairs_from(arr):
    return arr.reshape(-1, 2).astype(np.int64)


def test_policy_classes_end_to_end(tmp_path, oracle):
    W, H, ncams, maxkp = 320, 240, 3, 6000
    exe = build_driver(str(tmp_path / "policy_driver"))
    imgs = []
    for c in range(ncams):
        img = synth.rect_image(W, H, n_rect=150, seed=1000 + (c % 2), noise_sigma=2.0 + c)   # cams 0 and 2 share content
        imgs.append(img)
        with open(tmp_path / ("img%d.pgm" % c), "wb") as f:
            f.write(b"P5\n# synthetic\n%d %d\n255\n" % (W, H))
            f.write(img.tobytes())
    out = subprocess.run([exe, str(tmp_path), str(ncams), str(W), str(H), str(maxkp)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    descs = []
    for c in range(ncams):
        pyr, want = oracle_detect(oracle, imgs[c])
        kps = np.fromfile(tmp_path / ("kps%d.bin" % c), dtype=synth.KP_DTYPE)
        assert same_kps(kps, want)
        d = np.fromfile(tmp_path / ("desc%d.bin" % c), dtype=np.uint8).reshape(-1, 64)
        assert np.array_equal(d, oracle.clatch(pyr, want))
        feat = np.fromfile(tmp_path / ("feat%d.bin" % c), dtype=np.float32).reshape(-1, 4)
        assert np.array_equal(feat, oracle.features_from_kps(want))
        descs.append(d)
    n_with_matches = 0
    for i in range(ncams):
        for j in range(i + 1, ncams):
            m = oracle.k2nn(descs[i], descs[j], 40)
            want = np.stack([np.nonzero(m >= 0)[0], m[m >= 0]], 1)
            path = tmp_path / ("pair_%d_%d.bin" % (i, j))
            if len(want) == 0:
                assert not path.exists()          # empty results are not inserted (GPUMatcher.hpp:150)
                continue
            n_with_matches += 1
            assert np.array_equal(_pairs_from(np.fromfile(path, dtype=np.uint32)), want)   # (i_ = query, j_ = train)
    assert n_with_matches >= 1
    m = oracle.k2nn(descs[0], descs[1], 40)
    assert np.array_equal(_pairs_from(np.fromfile(tmp_path / "single_0_1.bin", dtype=np.uint32)),
                          np.stack([np.nonzero(m >= 0)[0], m[m >= 0]], 1))
    # map tracking: train = map (camera 0), query = camera 1, thr 60 -> IndMatch(map idx, query idx)
    m = oracle.k2nn(descs[1], descs[0], 60)
    assert np.array_equal(_pairs_from(np.fromfile(tmp_path / "map_1.bin", dtype=np.uint32)),
                          np.stack([m[m >= 0], np.nonzero(m >= 0)[0]], 1))
    # map <-> map: Q = map1, T = map2, thr 60 -> IndMatch(map1 idx, map2 idx)
    m = oracle.k2nn(descs[0], descs[1], 60)
    assert np.array_equal(_pairs_from(np.fromfile(tmp_path / "mapmap_0_1.bin", dtype=np.uint32)),
                          np.stack([np.nonzero(m >= 0)[0], m[m >= 0]], 1))
    # ---- regions edited in place after the detector published them, policy classes' DEFAULT mode: every entry
    # answers for the rows the block holds now
    def pairs_of(m):
        return np.stack([np.nonzero(m >= 0)[0], m[m >= 0]], 1)
    assert np.array_equal(_pairs_from(np.fromfile(tmp_path / "edit0_0_1.bin", dtype=np.uint32)), pairs_of(oracle.k2nn(descs[0], descs[1], 40)))
    e1 = np.fromfile(tmp_path / "edit1_desc0.bin", dtype=np.uint8).reshape(-1, 64)
    assert e1.shape == descs[0].shape and (e1 != descs[0]).any(axis=1).sum() == 1
    assert np.array_equal(_pairs_from(np.fromfile(tmp_path / "edit1_0_1.bin", dtype=np.uint32)), pairs_of(oracle.k2nn(e1, descs[1], 40)))
    e2 = np.fromfile(tmp_path / "edit2_desc0.bin", dtype=np.uint8).reshape(-1, 64)
    assert (e2 != descs[0]).any(axis=1).sum() > 20
    want = pairs_of(oracle.k2nn(e2, descs[1], 40))
    assert len(want) > len(pairs_of(oracle.k2nn(descs[0], descs[1], 40)))           # the rows copied over from camera 1 match it now
    assert np.array_equal(_pairs_from(np.fromfile(tmp_path / "edit2_0_1.bin", dtype=np.uint32)), want)
    m = oracle.k2nn(e2, descs[0], 60)                                                # query = edited block, map = camera 0 as detected
    assert np.array_equal(_pairs_from(np.fromfile(tmp_path / "edit2_map_0.bin", dtype=np.uint32)), np.stack([m[m >= 0], np.nonzero(m >= 0)[0]], 1))
    assert np.array_equal(_pairs_from(np.fromfile(tmp_path / "edit2_mapmap_1_0.bin", dtype=np.uint32)), pairs_of(oracle.k2nn(descs[1], e2, 60)))
