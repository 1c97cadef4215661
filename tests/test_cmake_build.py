"""CMakeLists.txt (the reference builds `koral` and links it into the node through CMake, reference
CMakeLists.txt:30-38,92-94): the library configures, builds for gfx950 and installs in the CPU container (hipcc cross-compiles), the
installed .so exports every symbol include/coloc_hip.h declares, and a consumer project finds it with find_package(coloc_hip) and links
the policy-class driver against the imported target."""
import ctypes as C
import os
import shutil
import subprocess

import pytest

from test_abi import _declared

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_configure_build_install_and_consume(tmp_path):
    if not shutil.which("cmake") or not os.path.isdir("/opt/rocm/lib/cmake/hip-lang"):
        pytest.skip("cmake or the ROCm HIP language package is not available")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    build, prefix = tmp_path / "build", tmp_path / "prefix"
    run = lambda cmd: subprocess.run(cmd, check=True, capture_output=True, text=True)
    run(["cmake", "-S", ROOT, "-B", str(build)] + gen)
    run(["cmake", "--build", str(build), "-j", "8"])
    run(["cmake", "--install", str(build), "--prefix", str(prefix)])
    so = prefix / "lib" / "libcoloc_hip.so"
    assert so.exists() and (prefix / "include" / "coloc_hip.h").exists() and (prefix / "include" / "coloc_hip" / "HIPMatcher.hpp").exists()
    assert (prefix / "lib" / "cmake" / "coloc_hip" / "coloc_hipConfig.cmake").exists()
    lib = C.CDLL(str(so))
    for name in _declared():
        assert hasattr(lib, name), "missing export " + name
    assert lib.clc_abi_version() == 4
    # the consumer: find_package(coloc_hip) + target_link_libraries(... coloc_hip::coloc_hip)
    cons = tmp_path / "consumer"
    run(["cmake", "-S", os.path.join(ROOT, "tests", "cmake_consumer"), "-B", str(cons), "-DCMAKE_PREFIX_PATH=" + str(prefix),
         "-DCOLOC_HIP_TESTS_DIR=" + os.path.join(ROOT, "tests")] + gen)
    run(["cmake", "--build", str(cons)])
    assert (cons / "policy_driver").exists()
