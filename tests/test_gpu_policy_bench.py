"""The drop-in path timed from C++ (tests/host/bench_policy.cpp: HIPDetector -> HIPMatcher -> HIPLocalizer in ColoC::mainThread's order,
reference include/coloc/coloc.hpp:111-148, the spans the reference prints at :129-136, :161-164, :217-225) must give, frame after frame,
the results of the device-pointer path: descriptors / features of the GPU front end (== the oracle's on the same frame), the map
matches of clc_match_2nn_dev, a pose at the rendered camera.  The figures themselves are reported by bench.py (`policy_path`)."""
import json
import os
import subprocess

import numpy as np
import pytest

import synth
from test_policy_host import build_driver
from test_gpu_detect import oracle_detect

pytestmark = pytest.mark.gpu

W, H = 640, 480
K = np.array([[520.0, 0, 320.0], [0, 520.0, 240.0], [0, 0, 1.0]])
PPU = 100.0


def render_scene(dirname, n_rect=900):
    """Three views of the textured relief: the keyframe the map is made of (cam_map) and the two cameras that track it."""
    tex = synth.plane_texture(n_rect=n_rect)
    relief = synth.smooth_relief()
    Ra, ta = synth.look_at_plane_pose((7.0, 7.0), 5.0, yaw=0.0, tilt=(0.10, -0.06))
    Rb, tb = synth.look_at_plane_pose((7.6, 6.7), 5.2, yaw=0.12, tilt=(-0.08, 0.09))
    Rc, tc = synth.look_at_plane_pose((6.5, 7.3), 4.9, yaw=-0.10, tilt=(0.05, 0.07))
    imgs = [synth.render_plane(tex, PPU, K, R, t, W, H, relief=relief) for R, t in ((Rc, tc), (Rb, tb), (Ra, ta))]
    for name, img in zip(("cam0", "cam1", "cam_map"), imgs):
        with open(os.path.join(dirname, name + ".pgm"), "wb") as f:
            f.write(b"P5\n# rendered\n%d %d\n255\n" % (W, H))
            f.write(img.tobytes())
    return imgs, relief, (Ra, ta), (Rb, tb)


def run_policy_bench(exe, dirname, frames, warmup, maxkp=12000, env=None):
    args = [exe, dirname, str(W), str(H), str(K[0, 0]), str(K[0, 2]), str(K[1, 2]), str(frames), str(warmup), str(maxkp)]
    r = subprocess.run(args, capture_output=True, text=True, cwd=dirname, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("POLICY ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(line[0][7:])


def test_policy_path_gives_the_device_path_results(tmp_path, oracle):
    from coloc_amd import Context
    exe = build_driver(str(tmp_path / "bench_policy"), "bench_policy.cpp")
    imgs, relief, (Ra, ta), (Rb, tb) = render_scene(str(tmp_path))
    # the map: the oracle's detector on the keyframe (== what HIPDetector finds), back-projected
    pyr0, kps0 = oracle_detect(oracle, imgs[2])
    feat0 = oracle.features_from_kps(kps0)
    synth.backproject_to_plane(feat0[:, :2].astype(np.float64), K, Ra, ta, relief=relief).astype(np.float64).tofile(tmp_path / "map_xyz.bin")
    for env_publish in (None, "0", "t"):
        env = dict(os.environ)
        if env_publish is not None:
            env["BENCH_POLICY_PUBLISH"] = env_publish
        out = run_policy_bench(exe, str(tmp_path), frames=24, warmup=6, env=env)
        assert out["failures"] == 0 and out["same_results_every_frame"] is True
        assert out["map_points"] == len(kps0)
        descs = []
        for c in range(2):
            pyr, kps = oracle_detect(oracle, imgs[c])
            d = np.fromfile(tmp_path / ("policy_desc%d.bin" % c), dtype=np.uint8).reshape(-1, 64)
            assert np.array_equal(d, oracle.clatch(pyr, kps)), (env_publish, c)
            f = np.fromfile(tmp_path / ("policy_kps%d.bin" % c), dtype=np.float32).reshape(-1, 4)
            assert np.array_equal(f, oracle.features_from_kps(kps)), (env_publish, c)
            descs.append(d)
        # map tracking: IndMatch(map idx, query idx), thr 60; the pair match of initMap: thr 40
        for c in range(2):
            m = oracle.k2nn(descs[c], oracle.clatch(pyr0, kps0), 60)
            assert out["map_matches"][c] == int((m >= 0).sum()), (env_publish, c)
        assert out["pair_matches"] == int((oracle.k2nn(descs[0], descs[1], 40) >= 0).sum())
        assert min(out["map_matches"]) > 60 and all(i > 0.6 * m for i, m in zip(out["pose_inliers"], out["map_matches"]))
        assert out["detect_us"] > 0 and out["match_us"] > 0 and out["pose_us"] > 0 and out["frame_us"] >= out["detect_us"]
        print("policy path (publish=%s): detect %.0f  match %.0f  pose %.0f  frame %.0f us, pair match %.0f us; %s keypoints"
              % (env_publish, out["detect_us"], out["match_us"], out["pose_us"], out["frame_us"], out["pair_match_us"], out["keypoints"]))
