"""Build libcoloc_hip.so (hipcc, gfx950 only) in-tree at coloc_amd/lib/.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels with the working tree to the GPU box.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libcoloc_hip.so")
SOURCES = ["capi.hip", "k2nn.hip", "clatch.hip", "lerp.hip", "pnp.hip", "detect.hip", "acransac.hip", "multicam.hip"]
HEADERS = ["clc_internal.h", "clc_sincos.h", "clc_acr.h", "p3p.h", "fivept.h", "fivept_wave.h", "latch_pattern.inc", "latch_layout.inc", os.path.join("..", "host", "HIPCovIntersection.hpp"), os.path.join("..", "..", "include", "coloc_hip.h")]
# -ffp-contract=off: the fp32 sample-coordinate / bilinear expressions and the fp64 residuals must
# evaluate in source order without fused multiply-add.
# -amdgpu-mfma-vgpr-form: MFMA accumulators in plain VGPRs (gfx950's register file is unified), so the K2NN top-2 reads
# them directly instead of through v_accvgpr_read copies.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-ldl"]


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into coloc_amd/lib/libcoloc_hip.so."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # CLC_EXTRA_FLAGS: extra compiler flags for experiments (e.g. "-DCLC_K2NN_AHEAD=2"); not set in any shipped build
    cmd = [hipcc_path()] + FLAGS + os.environ.get("CLC_EXTRA_FLAGS", "").split() + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
