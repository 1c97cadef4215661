"""bench.py contract checks on the GPU box: the default N=1 line has every required field, and the N>1
code path (one camera per rank, all-gather, sharded pair jobs, max-over-ranks timing) runs end to end in
REHEARSAL mode (2 ranks sharing the one GPU, gloo all-gather staged through the host -- the RCCL run
itself is the driver's, on a multi-GPU node)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    for line in reversed(out.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in output:\n" + out[-2000:])


def test_default_line_has_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.1 < r["frac"] < 1.0
    assert r["traffic"] is None or "traffic_source" in r
    assert 1.0 < r["clock_ghz_in_kernel"]["median"] <= 2.5          # the chip's real in-kernel clock, not an assumed 2.4
    assert d["sustained"]["seconds"] >= 1.0 and d["sustained"]["steps"] >= 20 and d["sustained_value"] > 0
    assert d["stages"]["k2nn_other_formulation"]["identical_results"] is True
    # round 5: the headline loop deals consecutive steps to lanes (contexts / streams) in turn, so that a step's sweep runs beside the next
    # steps' describe; the one-stream loop (rounds 1-4's headline, where `roofline` is measured) is reported beside it: same matches, not faster
    assert "LANES" in d["launch_mode"] and d["pipelined"]["lanes_identical_results"] is True and d["pipelined"]["lanes"] == 3
    one = d["one_stream"]
    assert one["identical_results"] is True and d["ms_per_step"] < 1.02 * one["ms_per_step"] and one["steps"] == d["steps"]
    assert "one-stream" in r["measured_in"] and d["pipelined"]["sweep_us_while_overlapped"] >= 0.9 * r["avg_launch_us"]
    assert d["accepted_matches_per_step"] > 5000                     # the two cameras see the same scene
    assert "acransac" in d["pose_solve"]["rule"].lower() or "a-contrario" in d["pose_solve"]["rule"]
    assert d["pose_solve_p50_ms"] > 0 and "section_errors" not in d
    assert 0 < d["pose_solve_p50_ms_c_abi"] <= d["pose_solve_p50_ms"] * 1.2      # the same solve without the wrapper's allocations
    # config[2]'s batched pose: the cameras' solves in one call give the single solves' results and cost less per pose than one after the other
    for k in ("cameras_4", "cameras_8"):
        pb = d["pose_batch"][k]
        assert pb["identical_results"] is True and 0 < pb["batch_p50_ms"] < pb["one_after_the_other_p50_ms"]
    assert d["settle"]["steps"] == 1000                              # the clock-settling steps are reported, not hidden
    assert d["warmup_effective"] == 3 + 20 + 1000 + 3 and 0 < d["value_cold"] <= 1.05 * d["value"]     # and so is the cold figure
    fe = d["front_end"]                                              # the rebuilt detector: two launches, <= 25 us per 640x480 frame
    assert fe["detect_us"] < 25.0 and fe["batch_of_8_us_per_camera"] < fe["frame_us_no_events"]
    assert d["stages"]["config2"]["real_front_end"]["Mmatches_per_s"] > 0
    hp = d["host_path"]             # a trusting context skips both uploads; a verifying one pays a pass over each host block instead
    assert hp["match_2nn_10k_x_10k_published_blocks_us"] < hp["match_2nn_10k_x_10k_us"] and hp["match_2nn_10k_x_10k_published_blocks_verified_us"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["openmvg_ratio_rule"]["value"] > 0
    assert d["value"] > 10 * c["value"]          # north-star target: >= 10x the host-CPU matcher


def test_one_stream_flag_gives_the_old_headline_loop():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--one-stream", "--headline-only",
                          "--no-cpu-baseline", "--sustain-seconds", "0.3"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["launch_mode"] == "eager launches, one stream" and d["one_stream"] is None and d["pipelined"] is None
    assert d["roofline"]["bound"] == "mfma" and "measured_in" not in d["roofline"] and d["value"] > 0


def _check_two_rank_line(d):
    assert d["n_gpus"] == 2 and d["config"]["pairs"] == 1 and d["value"] > 0 and "REHEARSAL" in d["collective"]
    # the guarded legs that run the same step through the C multi-camera entry points (clc-rccl / clc-peer at N > 1; rehearsal handles
    # here: the ranks share the one GPU) went through their whole control flow and reproduced the headline exchange's matches
    legs = d["exchange_legs"]
    assert set(legs) == {"clc-rccl", "clc-peer", "clc-rccl-overlap"}
    for name, leg in legs.items():
        assert leg["identical"] is True and leg["us_per_step"] > 0 and "error" not in leg and leg["steps"] == d["steps"]
        # round 6: every leg records what RCCL says about each rank's communicator (rehearsal handles have none: [0, -1]) and whether
        # its steps were overlapped (step k + 1's describe + exchange beside step k's sweep: same matches)
        assert leg["rccl_ranks"] == [[0, -1], [0, -1]] and leg["overlapped_steps"] is (name == "clc-rccl-overlap")
    assert d["value_cold"] > 0 and d["warmup_effective"] >= d["warmup"]
    # round 5: the product's exchange, having agreed with the torch step on both ranks, IS the headline: its K timed steps give value /
    # ms_per_step, the torch exchange's figure stays beside it, nothing fell back; and the line says what an efficiency against N = 1 means
    assert d["collective"].startswith("clc_mc_gather_enqueue_dev") and d["collective_fallback"] is None
    assert abs(d["ms_per_step"] - legs["clc-rccl"]["us_per_step"] / 1e3) < 1e-9 and d["torch_exchange"]["ms_per_step"] > 0
    assert abs(d["value"] - 1e8 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert "E(2)" in d["efficiency_note"] and "section_errors" not in d


def test_gpus_2_typed_directly_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` (the form of the driver's N = 1 command, no launcher in front) used to raise
    SystemExit.  It now starts `python -m torch.distributed.run` as a child before anything touches the GPU, relays its output and
    exit code -- and the N = 2 line comes out (rehearsal: two ranks share the one GPU)."""
    env = dict(os.environ, BENCH_FAULT_AFTER="200")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "4", "--settle-steps", "6"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "starting the ranks myself" in out.stderr
    _check_two_rank_line(_last_json(out.stdout))
    # and the RCCL form on a host with fewer GPUs than ranks says so instead of failing somewhere inside (one-GPU boxes only)
    import torch
    if torch.cuda.device_count() < 2:
        bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--settle-steps", "2"],
                             capture_output=True, text=True, cwd=ROOT, env=env, timeout=280)
        assert bad.returncode != 0 and "needs 2 GPUs" in (bad.stdout + bad.stderr)


def test_two_rank_rehearsal_runs():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BENCH_FAULT_AFTER="200")   # a hung rank dumps its stacks and exits
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
           "--settle-steps", "6", "--backend", "gloo"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    _check_two_rank_line(_last_json(out.stdout))
    # --overlap-steps: the overlapped form of the product's exchange is promoted instead (default off until a multi-GPU node measured it)
    out = subprocess.run(cmd + ["--overlap-steps"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    d = _last_json(out.stdout)
    assert d["collective_fallback"] is None and abs(d["ms_per_step"] - d["exchange_legs"]["clc-rccl-overlap"]["us_per_step"] / 1e3) < 1e-9
    assert d["exchange_legs"]["clc-rccl-overlap"]["identical"] is True
    # --exchange torch: never promote -- the torch exchange is the headline, the legs stay informational
    out = subprocess.run(cmd + ["--exchange", "torch"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    d = _last_json(out.stdout)
    assert d["collective"].startswith("REHEARSAL: gloo") and d["torch_exchange"] is None and d["collective_fallback"] is None
    assert all(leg["identical"] is True for leg in d["exchange_legs"].values())


def test_four_rank_rehearsal_runs():
    """Four cameras on four ranks (sharing the one GPU, gloo staging): 6 pairs dealt over 4 ranks, every rank takes part in
    the same number of collectives through warm-up, timed, sustained and breakdown legs."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BENCH_FAULT_AFTER="200")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "4", "--warmup", "1",
           "--settle-steps", "6", "--backend", "gloo", "--sustain-seconds", "0.2"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=280)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 4 and d["config"]["pairs"] == 6 and d["value"] > 0 and "REHEARSAL" in d["collective"]
    assert all(leg["identical"] is True for leg in d["exchange_legs"].values())


def test_streaming_scenario_runs_and_localizes():
    """BASELINE config[4] shape on one GPU, on RENDERED frames: the descriptors the front end computes are the ones matched
    against the map (CLATCH output of a reference view) and the a-contrario pose must land on the pose each frame was
    rendered from -- every frame localized, sub-percent position error."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_stream.py"), "--cams", "2", "--frames", "6"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["localized_frames"] == 12 and d["keypoints_p50"] > 3000 and d["map_matches_p50"] > 300
    assert d["inliers_p50"] > 0.8 * d["map_matches_p50"]
    assert d["position_error_p50"] < 0.005 * d["camera_height"] and d["position_error_max"] < 0.02 * d["camera_height"]
    assert d["cameras_at_30fps_per_gpu"] > 8
    # the inter-camera step (coloc.hpp:274-392) is the real chain now -- frame-to-frame match, a-contrario five-point, relative
    # pose from E, scale from the shared map features, refinement, covariance intersection -- with nothing taken from the rendered
    # poses: every pair must go through, the neighbour-derived position must land within 2 % of the camera height (median) and
    # the fused position must stay as good as the camera's own estimate
    assert d["inter_steps"] == 12 and d["inter_failures"] == 0
    assert d["pair_matches_p50"] > 500 and d["pair_inliers_p50"] > 0.8 * d["pair_matches_p50"] and d["common_map_features_p50"] > 50
    assert d["position_error_inter_p50"] < 0.02 * d["camera_height"]
    assert d["position_error_fused_p50"] < 0.005 * d["camera_height"] and d["position_error_fused_max"] < 0.02 * d["camera_height"]
    _check_inter_forms(d, 12)


def _check_inter_forms(d, steps):
    """Round 6: the inter-camera step finds the features the pair's temporary map shares with the global map the reference's way (the
    headline: temporary-map descriptors matched against the map's, coloc.hpp:317-323) and, in a second run of the same loop, through the
    source frame's map indices (rounds 3-5); both forms are reported and must land on the same positions to the scale rule's noise."""
    f = d["inter_forms"]
    ref, short = f["reference_chain"], f["map_index_shortcut"]
    assert ref["inter_steps"] == steps and short["inter_steps"] == steps and ref["inter_failures"] == 0 and short["inter_failures"] == 0
    assert ref["camera_frames_per_s_incl_inter"] == d["camera_frames_per_s_incl_inter"] and short["camera_frames_per_s_incl_inter"] > 0
    assert ref["common_map_features_p50"] > 50 and short["common_map_features_p50"] > 50
    assert f["centres_compared"] == steps and f["centre_difference_p50"] < 0.01 * d["camera_height"]
    assert ref["position_error_inter_p50"] < 0.02 * d["camera_height"] and short["position_error_inter_p50"] < 0.02 * d["camera_height"]


def test_streaming_scenario_eight_cameras():
    """BASELINE config[4]'s camera count on one GPU: 8 cameras x 4 frames through the frame-batched loop (one front-end call, one counted
    map-match launch, one batched a-contrario solve and ONE clc_inter_pose_batch call of eight pairs per frame: the lockstep form of the
    two-view filters), every frame localized, every pair through, both forms of the inter-camera step."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_stream.py"), "--cams", "8", "--frames", "4"],
                         capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["localized_frames"] == 32 and d["inter_steps"] == 32 and d["inter_failures"] == 0 and d["same_poses_both_modes"] is True
    assert d["position_error_p50"] < 0.005 * d["camera_height"] and d["position_error_inter_p50"] < 0.02 * d["camera_height"]
    assert d["position_error_fused_p50"] < 0.005 * d["camera_height"]
    _check_inter_forms(d, 32)


def test_streaming_scenario_synthetic_descriptors():
    """The round-1 harness (frames without geometry, observed descriptors synthesised from the map) still runs."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_stream.py"), "--scene", "synthetic", "--cams", "2", "--frames", "6",
                          "--map-points", "2000"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["position_error_p50"] < 0.01 and d["inliers_p50"] > 500


def test_streaming_scenario_two_lanes_same_poses():
    """Four cameras on the one GPU: the two-lane loop (front end + counted map match of a lane's next camera frame enqueued without a
    host synchronisation before its current pose solve, two host threads) finds bit for bit the poses of the camera-by-camera loop,
    and the inter-camera step goes through for every pair."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_stream.py"), "--cams", "4", "--frames", "6"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    # round 4: the headline loop is the frame-batched one (one front-end call, one counted match launch, ONE batched a-contrario solve per
    # frame); the two-lane loop still runs beside it and all three call patterns must find bit for bit the same poses
    assert d["mode"].startswith("frame-batched") and d["same_poses_both_modes"] is True
    assert "pipelined" in d["frame_p50_ms"] and "batched" in d["frame_p50_ms"] and d["two_lanes"]["cameras_at_30fps_per_gpu"] > 8
    assert d["localized_frames"] == 24 and d["sequential"]["localized_frames"] == 24
    assert d["inter_steps"] == 24 and d["inter_failures"] == 0
    assert d["position_error_p50"] < 0.005 * d["camera_height"] and d["position_error_fused_p50"] < 0.005 * d["camera_height"]
    assert d["frame_p50_ms"]["pipelined"] > 0 and d["frame_p50_ms"]["sequential"] > 0
