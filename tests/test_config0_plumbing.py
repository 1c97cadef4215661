"""BASELINE config[0]: "2 synthetic 640x480 frames, OpenMVG AKAZE + CPU brute-force Hamming matcher (plumbing, no GPU)".
OpenMVG (AKAZE, DistanceRatioMatch) is an empty submodule in the reference tree, so -- as SURVEY.md section 7 R5 lays
down -- the plumbing run uses the ORACLE describer (FAST-9 + orientation + CLATCH restatements) behind the same 64-byte
binary-regions layout, and the restated CPU matcher with OpenMVG's distance-ratio rule (reference
include/coloc/CPUMatcher.hpp:67-76: DistanceRatioMatch(0.8, BRUTE_FORCE_HAMMING, regions[first], regions[second])).
CPU only; no part of the product is involved.  What it pins down: the data that flows between the detector and matcher
policy classes (features {s x, s y, 7 s, angle}, 64 raw descriptor bytes, IndMatch pairs) for two frames end to end."""
import numpy as np

import synth
from test_gpu_detect import oracle_detect


def _frame(c):
    scene = synth.rect_image(640, 480, seed=1000, noise_sigma=0.0).astype(np.float32)
    rng = np.random.default_rng(1100 + c)
    return np.clip(scene + rng.normal(0.0, 2.0, scene.shape) + 0.5, 0, 255).astype(np.uint8)


def test_two_frames_through_the_cpu_detector_and_matcher(oracle):
    regions = {}
    for c in range(2):                                           # FeatureDetector::detectFeaturesFile per frame
        pyr, kps = oracle_detect(oracle, _frame(c))
        desc = oracle.clatch(pyr, kps)                           # Binary_Regions<SIOPointFeature, 64>::Descriptors()
        feat = oracle.features_from_kps(kps)                     # ...::Features(): {s x, s y, 7 s, angle}, GPUDetector.hpp:172-179
        assert desc.shape == (len(kps), 64) and feat.shape == (len(kps), 4) and len(kps) > 2000
        assert np.allclose(feat[:, 2], 7.0 * np.float32(1.2) ** kps["scale"], rtol=1e-6)
        regions[c] = dict(kps=kps, desc=desc, feat=feat)
    # FeatureMatcher::computeMatches over handlePairs(2) = {(0, 1)}: query = regions[0], database = regions[1]
    q, t = regions[0]["desc"], regions[1]["desc"]
    m_ratio, nthr = oracle.k2nn_omp(q, t, rule=1, ratio=0.8)     # DistanceRatioMatch(0.8): best < 0.8^2 * second
    m_k2nn = oracle.k2nn(q, t, 40)                               # the GPU path's acceptance rule on the same data
    assert nthr >= 1 and m_ratio.shape == (len(q),)
    ind = [(i, int(j)) for i, j in enumerate(m_ratio) if j >= 0]  # IndMatch(i_, j_), ascending i_
    assert len(ind) > 0.5 * min(len(q), len(t))
    # the two frames are the same scene under independent sensor noise: a match must land on (almost) the same place
    d = np.array([np.abs(regions[0]["feat"][i, :2] - regions[1]["feat"][j, :2]).max() for i, j in ind])
    assert np.mean(d <= 1.5 * 1.2 ** 7) > 0.98
    # the two acceptance rules agree on which train row is nearest wherever both accept
    both = (m_ratio >= 0) & (m_k2nn >= 0)
    assert both.sum() > 1000 and np.array_equal(m_ratio[both], m_k2nn[both])
