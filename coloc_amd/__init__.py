"""coloc_amd -- MI355X (gfx950) implementation of CoLoC's describe -> match -> pose-scoring hot path.

The product is the C-ABI shared library (include/coloc_hip.h, coloc_amd/lib/libcoloc_hip.so) plus
the C++ policy classes in coloc_amd/host/ that mirror the reference's GPUDetector / GPUMatcher.
This Python package is the thin binding used by the tests, bench.py and the one-camera-per-GPU
orchestration (torch.distributed is plumbing only).  There is NO CPU fallback: loading fails
loudly when the HIP library is missing.
"""
from .abi import (CLCError, Context, DetectorOptions, MatcherOptions, KP_DTYPE, lib_path, load_library,  # noqa: F401
                  keypoints_to_features, cov_intersection, MultiCam, mc_plan)

__all__ = ["CLCError", "Context", "DetectorOptions", "MatcherOptions", "KP_DTYPE", "lib_path",
           "load_library", "keypoints_to_features", "cov_intersection", "MultiCam", "mc_plan"]
