"""bench.py's launcher logic without a GPU (round 5): `--gpus N` typed without a launcher must start its own ranks as a CHILD process
before anything imports torch or touches the GPU, pass the same arguments on, relay output and exit code.  (What the ranks then measure
needs a GPU: tests/test_gpu_bench.py.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_world_size_spawns_torchrun_child():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["COLOC_HIP_LIB"] = "/nonexistent/libcoloc_hip.so"          # the ranks stop at the first thing they need: no measurement on the CPU
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--settle-steps", "0"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=240)
    assert "starting the ranks myself" in out.stderr
    assert "torch.distributed.run" in out.stderr and "--nproc-per-node 2" in out.stderr and "--master-addr 127.0.0.1" in out.stderr
    assert "--gpus 2 --backend gloo --steps 1" in out.stderr                   # the child gets the same arguments
    assert out.returncode != 0                                                  # ... and its failure is the parent's exit code


def test_under_a_launcher_nothing_is_spawned():
    # WORLD_SIZE set (what torch.distributed.run exports): bench.py must NOT start another launcher; with a world that does not match
    # --gpus it says so and leaves
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=240)
    assert "starting the ranks myself" not in out.stderr
    assert out.returncode != 0 and "WORLD_SIZE=3" in (out.stdout + out.stderr)
