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
SOURCES = ["capi_core.hip", "capi_match.hip", "desc_cache.hip", "capi_pose.hip", "pose_batch.hip", "inter_pose.hip", "inter_geometry.cpp",
           "k2nn.hip", "clatch.hip", "lerp.hip", "pnp.hip", "detect.hip", "acransac.hip", "multicam.hip"]
HEADERS = ["clc_internal.h", "clc_ctx.h", "desc_cache.h", "inter_geometry.h", "clc_sincos.h", "clc_acr.h", "p3p.h", "fivept.h", "fivept_wave.h", "twoview_min.h", "latch_pattern.inc", "latch_layout.inc", "latch_layout_swap.inc", os.path.join("..", "host", "HIPCovIntersection.hpp"), os.path.join("..", "host", "HIPRobustMatcher.hpp"), os.path.join("..", "host", "coloc_hip_geometry.hpp"), os.path.join("..", "..", "include", "coloc_hip.h")]
# -ffp-contract=off: the fp32 sample-coordinate / bilinear expressions and the fp64 residuals must
# evaluate in source order without fused multiply-add.
# -amdgpu-mfma-vgpr-form: MFMA accumulators in plain VGPRs (gfx950's register file is unified), so the K2NN top-2 reads
# them directly instead of through v_accvgpr_read copies.
CFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form",
          "-I", os.path.join(os.path.dirname(PKG), "include")]          # (inter_geometry.cpp includes host/HIPRobustMatcher.hpp, which names coloc_hip.h plainly)
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]


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
    missing = [d for d in deps if not os.path.exists(d)]
    if missing:
        raise RuntimeError("build.py lists files that do not exist: %s" % ", ".join(missing))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into coloc_amd/lib/libcoloc_hip.so."""
    if not force and not needs_build():
        return LIB
    # CLC_EXTRA_FLAGS: extra compiler flags for experiments (e.g. "-DCLC_K2NN_AHEAD=2"); not set in any shipped build (their objects
    # are kept apart from the shipped build's)
    extra = os.environ.get("CLC_EXTRA_FLAGS", "").split()
    objdir = os.path.join(LIBDIR, "obj_extra" if extra else "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = hipcc_path()
    # one translation unit per process, all of them at once (the sources are independent; a single hipcc call compiles them one after
    # the other: 25 s against 10 s on the 8 cores of the build container), then one link
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    hdr_time = max(hdr_time, os.path.getmtime(os.path.abspath(__file__)))
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        stale = force or extra or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(os.path.join(CSRC, src)), hdr_time)
        if stale:
            # (.cpp: host arithmetic only, compiled as plain C++ by the same driver)
            flags = [f for f in CFLAGS if not f.startswith("--offload-arch") and f not in ("-mllvm", "-amdgpu-mfma-vgpr-form")] if src.endswith(".cpp") else CFLAGS
            jobs.append([hipcc] + flags + extra + (["-x", "c++"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj])
    if verbose:
        for j in jobs:
            print(" ".join(j))
    procs = [subprocess.Popen(j, cwd=CSRC) for j in jobs]
    failed = [j for j, p in zip(jobs, procs) if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    link = [hipcc] + LDFLAGS + ["-o", LIB] + [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
