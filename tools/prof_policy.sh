#!/bin/bash
# Kernel trace of the drop-in path (tests/host/bench_policy.cpp).  Run ON the GPU box from the repo root:
#   tools/prof_policy.sh <tag> [frames]          -> gpurun_out/prof_policy_<tag>/ (rocprofv3 --kernel-trace --stats)
set -e
tag=${1:-r06}; frames=${2:-300}
out=gpurun_out/policy_$tag
mkdir -p $out
python - <<PY
import os, sys, subprocess
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth
from test_gpu_policy_bench import W, H, K, PPU
from coloc_amd import Context, keypoints_to_features
d = "$out"
relief = synth.smooth_relief()
poses = [synth.look_at_plane_pose((6.5, 7.3), 4.9, yaw=-0.10, tilt=(0.05, 0.07)), synth.look_at_plane_pose((7.6, 6.7), 5.2, yaw=0.12, tilt=(-0.08, 0.09)),
         synth.look_at_plane_pose((7.0, 7.0), 5.0, yaw=0.0, tilt=(0.10, -0.06))]
tex = synth.plane_texture(n_rect=int(os.environ.get("POLICY_NRECT", "2500")))
frames = [synth.render_plane(tex, PPU, K, R, t, W, H, relief=relief) for R, t in poses]
for name, img in zip(("cam0", "cam1", "cam_map"), frames):
    with open(os.path.join(d, name + ".pgm"), "wb") as f:
        f.write(b"P5\n# rendered\n%d %d\n255\n" % (W, H)); f.write(img.tobytes())
det = Context(device=0, width=W, height=H, maxkp=12000, matcher=False)
kps, _, _ = det.detect_and_describe(frames[2]); det.close()
feat = keypoints_to_features(kps)
synth.backproject_to_plane(feat[:, :2].astype(np.float64), K, *poses[2], relief=relief).astype(np.float64).tofile(os.path.join(d, "map_xyz.bin"))
lib = os.path.abspath("coloc_amd/lib")
subprocess.check_call(["g++", "-std=c++14", "-O2", "-I", "include", "-I", "coloc_amd/host", "tests/host/bench_policy.cpp", "-o", os.path.join(d, "bench_policy"),
                       "-L", lib, "-lcoloc_hip", "-Wl,-rpath," + lib])
PY
here=$(pwd)
export TMPDIR=/tmp
rm -rf $here/gpurun_out/prof_policy_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $here/gpurun_out/prof_policy_$tag -o policy -- $here/$out/bench_policy $here/$out 640 480 520 320 240 $frames 60 12000 > $out/run.log 2>&1 || true
grep POLICY $out/run.log || tail -5 $out/run.log
grep "front end" $out/run.log | tail -2 || true
find gpurun_out/prof_policy_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf gpurun_out/prof_policy_$tag $out/*.bin $out/*.pgm $out/bench_policy
cut -d, -f1-8 $out/kernel_stats.csv | head -16
