"""The C++ policy classes (coloc_amd/host/HIPDetector.hpp, HIPMatcher.hpp) compile and link against
libcoloc_hip.so with plain g++ (no HIP headers needed on the host side) -- CPU check; the run is in
tests/test_gpu_policy.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_driver(out, source="policy_driver.cpp"):
    from coloc_amd import build
    lib = build.build()
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "coloc_amd", "host"),
           os.path.join(ROOT, "tests", "host", source), "-o", out, "-L", os.path.dirname(lib), "-lcoloc_hip",
           "-Wl,-rpath," + os.path.dirname(lib)]
    subprocess.check_call(cmd)
    return out


def test_policy_classes_compile_and_link(tmp_path):
    exe = build_driver(str(tmp_path / "policy_driver"))
    assert os.path.exists(exe)


def test_localizer_and_robust_matcher_compile_and_link(tmp_path):
    exe = build_driver(str(tmp_path / "localizer_driver"), "localizer_driver.cpp")
    assert os.path.exists(exe)


def test_policy_bench_compiles_and_links(tmp_path):
    """tests/host/bench_policy.cpp (the drop-in path timed from C++; run by tests/test_gpu_policy_bench.py and bench.py)."""
    exe = build_driver(str(tmp_path / "bench_policy"), "bench_policy.cpp")
    assert os.path.exists(exe)
