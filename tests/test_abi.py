"""The C-ABI library loads on a machine without a GPU and exports every symbol include/coloc_hip.h
declares; struct layouts match the reference's wire formats.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "coloc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(clc_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from coloc_amd import abi
    lib = abi.load_library()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(abi.EXPORTS) == names
    assert lib.clc_abi_version() == abi.ABI_VERSION == 4
    assert lib.clc_status_string(2) == b"capacity exceeded"


def test_struct_layouts():
    from coloc_amd import abi
    assert abi.KP_DTYPE.itemsize == 20                       # Keypoint.h:155-163
    assert [abi.KP_DTYPE.fields[k][1] for k in ("x", "y", "score", "angle", "scale")] == [0, 4, 8, 12, 16]
    assert C.sizeof(abi.DetectorOptions) == 24 and C.sizeof(abi.MatcherOptions) == 12
    assert C.sizeof(abi.MatchJob) == 24


def test_missing_library_fails_loudly(monkeypatch):
    from coloc_amd import abi
    monkeypatch.setattr(abi, "_lib", None)
    monkeypatch.setenv("COLOC_HIP_LIB", "/nonexistent/libcoloc_hip.so")
    with pytest.raises(abi.CLCError):
        abi.load_library()


def test_no_device_is_an_error_code_not_a_crash():
    """In the CPU container there is no GPU: ctx_create must return CLC_ERR_NO_DEVICE (4)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from coloc_amd import Context, CLCError
    with pytest.raises(CLCError) as e:
        Context(device=0)
    assert e.value.status == 4


def test_keypoints_to_features_is_host_only(oracle):
    import synth
    from coloc_amd import keypoints_to_features
    kps = synth.random_keypoints(64, 640, 480, seed=1)
    assert np.array_equal(keypoints_to_features(kps), oracle.features_from_kps(kps))


def test_share_grain_is_a_multiple_of_every_sweep_workgroup():
    """multicam cuts rank shares on QBLOCK queries; a sweep workgroup covers clc_k2nn_queries_per_block() queries
    (both formulations; callable without a context, hence without a GPU)."""
    from coloc_amd import abi, multicam
    lib = abi.load_library()
    grain = lib.clc_k2nn_queries_per_block(None)
    assert grain in (128, 256) and multicam.QBLOCK % grain == 0 and multicam.QBLOCK % 128 == 0


def test_job_structs_match_the_c_header(tmp_path):
    """The batch entries take arrays of plain C structs (clc_pose_job, clc_two_view_job, clc_inter_pose_job): the ctypes mirrors in
    coloc_amd/abi.py must have the C compiler's sizes and field offsets (a gcc probe over include/coloc_hip.h, no GPU needed)."""
    import subprocess
    from coloc_amd import abi
    probes = {"clc_pose_job": (abi.PoseJob, ["X", "n", "seed", "precision", "refine", "huber_a", "Rt", "n_inliers", "error_max", "rmse"]),
              "clc_two_view_job": (abi.TwoViewJob, ["x1", "K2", "n", "max_iteration", "seed", "precision", "E", "inliers", "n_inliers", "status", "error_max", "min_nfa"]),
              "clc_inter_pose_job": (abi.InterPoseJob, ["tv", "map_index", "map_X", "map_n", "Rt_source", "huber_a", "d_first_desc", "first_feature", "d_map_desc",
                                                        "match_threshold", "Rt", "cov", "rmse", "scale", "n_front", "stage", "n_map_matches"]),
              "clc_desc_handle": (abi.DescHandle, ["host", "generation", "count", "slot"])}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "coloc_hip.h"', 'int main(void) {']
    for name, (_, fields) in probes.items():
        src.append('printf("%s %%zu", sizeof(%s));' % (name, name))
        for f in fields:
            src.append('printf(" %%zu", offsetof(%s, %s));' % (name, f))
        src.append('printf("\\n");')
    src += ['return 0;', '}']
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    for line in out.strip().splitlines():
        parts = line.split()
        cls, fields = probes[parts[0]]
        assert C.sizeof(cls) == int(parts[1]), parts[0]
        for f, off in zip(fields, parts[2:]):
            assert getattr(cls, f).offset == int(off), (parts[0], f)
