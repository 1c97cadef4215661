"""ctypes binding of include/coloc_hip.h.  Thin: every method is one C-ABI call.

No fallback of any kind: if libcoloc_hip.so is missing or a call fails, CLCError is raised.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))

KP_DTYPE = np.dtype(
    {"names": ["x", "y", "score", "angle", "scale"],
     "formats": ["<i4", "<i4", "u1", "<f4", "u1"],
     "offsets": [0, 4, 8, 12, 16], "itemsize": 20})   # clc_keypoint == reference Keypoint.h:155-163

CLC_OK, CLC_ERR_BAD_ARG, CLC_ERR_CAPACITY, CLC_ERR_HIP, CLC_ERR_NO_DEVICE, CLC_ERR_STATE = range(6)


class CLCError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("coloc_hip status %d: %s" % (status, msg))
        self.status = status


class DetectorOptions(C.Structure):   # coloc::DetectorOptions, colocData.hpp:29-36
    _fields_ = [("scale_factor", C.c_float), ("scale_levels", C.c_uint8), ("width", C.c_uint32),
                ("height", C.c_uint32), ("maxkp", C.c_uint32), ("thresh", C.c_uint8)]


class MatcherOptions(C.Structure):    # coloc::MatcherOptions, colocData.hpp:38-42
    _fields_ = [("distRatio", C.c_float), ("thresh", C.c_int), ("maxkp", C.c_uint32)]


class McShare(C.Structure):           # clc_mc_share
    _fields_ = [("first", C.c_int32), ("second", C.c_int32), ("q_begin", C.c_uint32), ("nq", C.c_uint32), ("out_offset", C.c_uint32)]


class MatchJob(C.Structure):          # clc_match_job
    _fields_ = [("q_offset", C.c_uint32), ("nq", C.c_uint32), ("t_offset", C.c_uint32), ("nt", C.c_uint32),
                ("out_offset", C.c_uint32), ("threshold", C.c_uint32)]


class PoseJob(C.Structure):
    """clc_pose_job (include/coloc_hip.h)"""
    _fields_ = [("X", C.c_void_p), ("x", C.c_void_p), ("K", C.c_void_p), ("n", C.c_int), ("max_iteration", C.c_int), ("seed", C.c_uint64),
                ("precision", C.c_double), ("refine", C.c_int), ("huber_a", C.c_double),
                ("Rt", C.c_void_p), ("cov", C.c_void_p), ("inlier_mask", C.c_void_p), ("inliers", C.c_void_p),
                ("n_inliers", C.c_int), ("iterations", C.c_int), ("status", C.c_int), ("error_max", C.c_double), ("rmse", C.c_double)]


ABI_VERSION = 4          # CLC_ABI_VERSION of include/coloc_hip.h
DESC_CACHE_OFF, DESC_CACHE_VERIFY, DESC_CACHE_TRUST = 0, 1, 2

class DescHandle(C.Structure):
    """clc_desc_handle (include/coloc_hip.h): what a publication hands out"""
    _fields_ = [("host", C.c_void_p), ("generation", C.c_uint64), ("count", C.c_uint32), ("slot", C.c_uint32)]


class TwoViewJob(C.Structure):
    """clc_two_view_job (include/coloc_hip.h)"""
    _fields_ = [("x1", C.c_void_p), ("x2", C.c_void_p), ("K1", C.c_void_p), ("K2", C.c_void_p), ("n", C.c_int), ("img_w", C.c_int), ("img_h", C.c_int),
                ("max_iteration", C.c_int), ("seed", C.c_uint64), ("precision", C.c_double),
                ("E", C.c_void_p), ("F", C.c_void_p), ("inlier_mask", C.c_void_p), ("inliers", C.c_void_p),
                ("n_inliers", C.c_int), ("iterations", C.c_int), ("status", C.c_int), ("error_max", C.c_double), ("min_nfa", C.c_double)]


class InterPoseJob(C.Structure):
    """clc_inter_pose_job (include/coloc_hip.h)"""
    _fields_ = [("tv", TwoViewJob), ("map_index", C.c_void_p), ("map_X", C.c_void_p), ("map_n", C.c_int), ("Rt_source", C.c_void_p), ("huber_a", C.c_double),
                ("d_first_desc", C.c_void_p), ("first_feature", C.c_void_p), ("d_map_desc", C.c_void_p), ("match_threshold", C.c_int),
                ("Rt", C.c_double * 12), ("cov", C.c_double * 36), ("rmse", C.c_double), ("scale", C.c_double),
                ("n_front", C.c_int), ("n_common", C.c_int), ("n_refined", C.c_int), ("stage", C.c_int), ("n_map_matches", C.c_int)]


EXPORTS = [
    "clc_abi_version", "clc_status_string", "clc_ctx_create", "clc_ctx_destroy", "clc_last_error_string",
    "clc_sync", "clc_stream", "clc_pyramid_build", "clc_pyramid_build_dev", "clc_pyramid_level",
    "clc_pyramid_download", "clc_describe", "clc_describe_dev", "clc_keypoints_to_features",
    "clc_match_2nn", "clc_match_2nn_dev", "clc_match_jobs_dev", "clc_set_map", "clc_match_map",
    "clc_pnp_residuals", "clc_pnp_score", "clc_profile_enable", "clc_profile_reset", "clc_profile_read",
    "clc_kernel_name", "clc_detect", "clc_detect_dev", "clc_detect_buffers", "clc_describe_detected_dev",
    "clc_detect_and_describe", "clc_detect_batch_dev", "clc_desc_cache_publish", "clc_desc_cache_clear", "clc_desc_cache_stats", "clc_desc_cache_mode", "clc_desc_handle_live", "clc_detect_and_describe_view", "clc_detect_store_descriptors", "clc_describe_match_pair_dev", "clc_essential_acransac_batch", "clc_inter_pose_batch", "clc_pnp_localize_ac_batch", "clc_match_pairs", "clc_pnp_ransac", "clc_pnp_p3p", "clc_pnp_refine", "clc_pnp_localize", "clc_epipolar_residuals", "clc_epipolar_score", "clc_cov_intersection", "clc_essential_ransac",
    "clc_essential_fivepoint", "clc_describe_batch_dev", "clc_match_map_dev", "clc_k2nn_set_formulation",
    "clc_k2nn_queries_per_block", "clc_k2nn_plan_query", "clc_k2nn_device_info", "clc_pnp_acransac", "clc_pnp_localize_ac", "clc_essential_acransac", "clc_k2nn_clock_check", "clc_ctx_device", "clc_mc_plan", "clc_mc_unique_id",
    "clc_mc_create", "clc_mc_destroy", "clc_mc_last_error_string", "clc_mc_arena", "clc_mc_gather_dev", "clc_mc_match_dev", "clc_mc_virtual_put", "clc_mc_open_peers",
    "clc_mc_gather_enqueue_dev", "clc_mc_match_enqueue_dev", "clc_mc_counts", "clc_mc_set_overlap", "clc_mc_comm_info", "clc_match_jobs_counted_dev",
    "clc_two_view_acransac", "clc_two_view_acransac_batch", "clc_two_view_minimal",
]
KERNELS = ["pyramid_kernel", "clatch_kernel", "k2nn_sweep_kernel", "k2nn_merge_kernel", "pnp_residual_kernel",
           "pnp_score_kernel", "detect_kernels"]

_lib = None


def lib_path():
    return os.environ.get("COLOC_HIP_LIB", os.path.join(_PKG, "lib", "libcoloc_hip.so"))


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so / libhsa-runtime64.so.  Two HIP runtimes in one
    process cannot both own the GPU ("No HIP GPUs are available" in whichever initialises second), so
    when torch is installed its runtime is mapped first and libcoloc_hip.so's DT_NEEDED
    libamdhip64.so.7 resolves to it.  A pure C/C++ host simply uses the system ROCm runtime."""
    if os.environ.get("COLOC_HIP_SYSTEM_RUNTIME"):
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """dlopen libcoloc_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise CLCError(-1, "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    _share_torch_hip_runtime()
    lib = C.CDLL(path)
    # the version BEFORE any symbol an older library does not export is resolved (include/coloc_hip.h CLC_ABI_VERSION)
    got = lib.clc_abi_version() if hasattr(lib, "clc_abi_version") else 0
    if got != ABI_VERSION:
        raise CLCError(-1, "%s reports ABI version %d, this binding needs %d: rebuild it (python -c 'import __graft_entry__ as g; g.build()')"
                       % (path, got, ABI_VERSION))
    lib.clc_status_string.restype = C.c_char_p
    lib.clc_last_error_string.restype = C.c_char_p
    lib.clc_last_error_string.argtypes = [C.c_void_p]
    lib.clc_stream.restype = C.c_void_p
    lib.clc_stream.argtypes = [C.c_void_p]
    lib.clc_ctx_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.clc_ctx_destroy.argtypes = [C.c_void_p]
    lib.clc_sync.argtypes = [C.c_void_p]
    vp, ci, u32, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_size_t
    lib.clc_pyramid_build.argtypes = [vp, vp, u32, u32]
    lib.clc_pyramid_build_dev.argtypes = [vp, vp, u32, u32, sz, vp]
    lib.clc_pyramid_level.argtypes = [vp, ci, C.POINTER(u32), C.POINTER(u32), C.POINTER(sz), C.POINTER(vp)]
    lib.clc_pyramid_download.argtypes = [vp, ci, vp]
    lib.clc_describe.argtypes = [vp, vp, ci, vp]
    lib.clc_describe_dev.argtypes = [vp, vp, ci, vp, vp]
    lib.clc_describe_batch_dev.argtypes = [vp, ci, vp, C.c_uint32, C.c_uint32, C.c_size_t, vp, vp, vp, vp]
    lib.clc_detect_batch_dev.argtypes = [vp, ci, vp, C.c_uint32, C.c_uint32, C.c_size_t, vp, vp, vp, vp]
    lib.clc_desc_cache_publish.argtypes = [vp, vp, vp, ci, vp]
    lib.clc_desc_handle_live.argtypes = [vp]
    lib.clc_detect_and_describe_view.argtypes = [vp, vp, u32, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(ci), C.POINTER(ci)]
    lib.clc_detect_store_descriptors.argtypes = [vp, vp, ci, vp]
    lib.clc_desc_cache_clear.argtypes = []
    lib.clc_desc_cache_stats.argtypes = [vp, vp]
    lib.clc_desc_cache_mode.argtypes = [vp, ci]
    lib.clc_essential_acransac_batch.argtypes = [vp, vp, ci]
    lib.clc_inter_pose_batch.argtypes = [vp, vp, ci]
    lib.clc_describe_match_pair_dev.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_size_t, vp, vp, vp, ci, vp, vp]
    lib.clc_pnp_localize_ac_batch.argtypes = [vp, vp, ci]
    lib.clc_keypoints_to_features.argtypes = [vp, ci, vp]
    lib.clc_match_2nn.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp, vp]
    lib.clc_match_2nn_dev.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp]
    lib.clc_match_jobs_dev.argtypes = [vp, vp, vp, ci, vp, vp]
    lib.clc_match_pairs.argtypes = [vp, vp, vp, ci, vp, ci, ci, vp]
    lib.clc_set_map.argtypes = [vp, vp, ci]
    lib.clc_match_map_dev.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.clc_match_map.argtypes = [vp, vp, ci, ci, vp]
    lib.clc_pnp_residuals.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp]
    lib.clc_pnp_score.argtypes = [vp, vp, ci, vp, vp, ci, vp, C.c_double, vp, vp]
    lib.clc_detect.argtypes = [vp, vp, ci, C.POINTER(ci), C.POINTER(ci)]
    lib.clc_detect_dev.argtypes = [vp, vp]
    lib.clc_detect_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.clc_describe_detected_dev.argtypes = [vp, vp, vp]
    lib.clc_detect_and_describe.argtypes = [vp, vp, u32, u32, vp, vp, ci, C.POINTER(ci), C.POINTER(ci)]
    lib.clc_pnp_ransac.argtypes = [vp, vp, vp, ci, vp, vp, ci, C.c_uint64, C.c_double, vp, vp, C.POINTER(ci), C.POINTER(C.c_double)]
    lib.clc_pnp_p3p.argtypes = [vp, vp, vp, ci, vp, vp, ci, vp]
    lib.clc_essential_ransac.argtypes = [vp, vp, vp, ci, vp, vp, vp, ci, C.c_uint64, C.c_double, vp, vp, vp, C.POINTER(ci)]
    lib.clc_essential_fivepoint.argtypes = [vp, vp, vp, ci, vp, vp, vp, ci, vp]
    lib.clc_epipolar_residuals.argtypes = [vp, vp, ci, vp, vp, ci, vp]
    lib.clc_epipolar_score.argtypes = [vp, vp, ci, vp, vp, ci, C.c_double, vp, vp]
    lib.clc_pnp_localize.argtypes = [vp, vp, vp, ci, vp, vp, ci, C.c_uint64, C.c_double, C.c_double, vp, vp, vp, C.POINTER(ci),
                                     C.POINTER(C.c_double)]
    lib.clc_pnp_refine.argtypes = [vp, vp, vp, ci, vp, vp, vp, C.c_double, ci, vp, vp, C.POINTER(C.c_double), C.POINTER(ci)]
    lib.clc_profile_enable.argtypes = [vp, ci]
    lib.clc_profile_reset.argtypes = [vp]
    lib.clc_profile_read.argtypes = [vp, ci, C.POINTER(C.c_double), C.POINTER(ci)]
    lib.clc_kernel_name.restype = C.c_char_p
    dp, ip = C.POINTER(C.c_double), C.POINTER(ci)
    lib.clc_pnp_acransac.argtypes = [vp, vp, vp, ci, vp, ci, C.c_uint64, C.c_double, vp, vp, vp, ip, dp, dp, ip]
    lib.clc_pnp_localize_ac.argtypes = [vp, vp, vp, ci, vp, ci, C.c_uint64, C.c_double, C.c_double, vp, vp, vp, vp, ip, dp, dp]
    lib.clc_essential_acransac.argtypes = [vp, vp, vp, ci, vp, vp, ci, ci, ci, C.c_uint64, C.c_double, vp, vp, vp, vp, ip, dp, dp, ip]
    lib.clc_two_view_acransac.argtypes = [vp, ci, vp, vp, ci, vp, vp, ci, ci, ci, C.c_uint64, C.c_double, vp, vp, vp, vp, ip, dp, dp, ip]
    lib.clc_two_view_acransac_batch.argtypes = [vp, ci, vp, ci]
    lib.clc_two_view_minimal.argtypes = [vp, ci, vp, vp, ci, ci, ci, vp, ci, vp]
    lib.clc_mc_plan.argtypes = [vp, ci, ci, ci, ci, vp, ci, ip]
    lib.clc_mc_unique_id.argtypes = [vp]
    lib.clc_mc_create.argtypes = [vp, vp, ci, ci, ci, C.POINTER(vp)]
    lib.clc_mc_destroy.argtypes = [vp]
    lib.clc_mc_last_error_string.argtypes = [vp]
    lib.clc_mc_last_error_string.restype = C.c_char_p
    lib.clc_mc_arena.argtypes = [vp, C.POINTER(vp), ip, ip]
    lib.clc_mc_gather_dev.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.clc_mc_virtual_put.argtypes = [vp, ci, vp, ci, vp]
    lib.clc_mc_open_peers.argtypes = [vp, vp]
    lib.clc_mc_match_dev.argtypes = [vp, ci, vp, ci, vp, ci, ip, vp]
    lib.clc_mc_gather_enqueue_dev.argtypes = [vp, vp, ci, vp, ci, vp]
    lib.clc_mc_match_enqueue_dev.argtypes = [vp, ci, vp, ci, vp, ci, ip, vp]
    lib.clc_mc_counts.argtypes = [vp, vp, vp]
    lib.clc_mc_set_overlap.argtypes = [vp, ci]
    lib.clc_mc_comm_info.argtypes = [vp, ip, ip]
    lib.clc_match_jobs_counted_dev.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp]
    lib.clc_k2nn_clock_check.argtypes = [vp, vp, ci, vp, ci, vp, vp, dp, dp, dp, ip]
    lib.clc_k2nn_set_formulation.argtypes = [vp, ci]
    lib.clc_k2nn_queries_per_block.argtypes = [vp]
    lib.clc_k2nn_plan_query.argtypes = [vp, ci, ci, vp]
    lib.clc_k2nn_device_info.argtypes = [vp, vp, vp]
    _lib = lib
    return lib


def _p(a):
    # (a.ctypes.data_as(c_void_p) builds two ctypes objects per call, ~2.3 us; a pose solve passes seven pointers.  The caller keeps
    # the array alive across the call.)
    return None if a is None else C.c_void_p(a.ctypes.data)


def cov_intersection(CA, CB, ca, cb):
    """Host-side covariance intersection: returns (omega, fused 3x3 covariance, fused position)."""
    lib = load_library()
    CA = np.ascontiguousarray(CA, dtype=np.float64).reshape(9); CB = np.ascontiguousarray(CB, dtype=np.float64).reshape(9)
    ca = np.ascontiguousarray(ca, dtype=np.float64).reshape(3); cb = np.ascontiguousarray(cb, dtype=np.float64).reshape(3)
    om = C.c_double(); cov = np.zeros(9); pos = np.zeros(3)
    lib.clc_cov_intersection.argtypes = [C.c_void_p] * 4 + [C.POINTER(C.c_double), C.c_void_p, C.c_void_p]
    rc = lib.clc_cov_intersection(_p(CA), _p(CB), _p(ca), _p(cb), C.byref(om), _p(cov), _p(pos))
    if rc != CLC_OK:
        raise CLCError(rc, lib.clc_status_string(rc).decode())
    return om.value, cov.reshape(3, 3), pos


def _two_view_fill(tv, keep, x1, x2, K1, K2, img_wh, max_iteration, seed, precision):
    x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2); x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
    K1 = np.ascontiguousarray(K1, dtype=np.float64).reshape(9); K2 = np.ascontiguousarray(K2, dtype=np.float64).reshape(9)
    n = x1.shape[0]
    E, F = np.zeros(9), np.zeros(9)
    mask, inl = np.zeros(max(n, 1), dtype=np.uint8), np.zeros(max(n, 1), dtype=np.int32)
    keep.append((x1, x2, K1, K2, E, F, mask, inl))
    tv.x1, tv.x2, tv.K1, tv.K2, tv.n = x1.ctypes.data, x2.ctypes.data, K1.ctypes.data, K2.ctypes.data, n
    tv.img_w, tv.img_h, tv.max_iteration, tv.seed, tv.precision = int(img_wh[0]), int(img_wh[1]), int(max_iteration), int(seed), float(precision)
    tv.E, tv.F, tv.inlier_mask, tv.inliers = E.ctypes.data, F.ctypes.data, mask.ctypes.data, inl.ctypes.data
    return E, F, mask, inl


def _two_view_result(tv, E, F, mask, inl):
    ok = tv.n_inliers > 0
    return dict(E=E.reshape(3, 3).copy() if ok else None, F=F.reshape(3, 3).copy() if ok else None, mask=mask[:tv.n].astype(bool),
                inliers=inl[:tv.n_inliers].copy(), error_max=tv.error_max, min_nfa=tv.min_nfa, iterations=tv.iterations, status=tv.status)


def essential_acransac_batch(ctxs, problems, max_iteration=256, precision=float("inf")):
    """clc_essential_acransac_batch: problems = [(x1, x2, K1, K2, (w, h), seed), ...], one Context per problem; -> list of result dicts."""
    lib = load_library()
    n = len(problems)
    jobs = (TwoViewJob * n)()
    keep, outs = [], []
    for j, (x1, x2, K1, K2, wh, seed) in zip(jobs, problems):
        outs.append(_two_view_fill(j, keep, x1, x2, K1, K2, wh, max_iteration, seed, precision))
    hs = (C.c_void_p * n)(*[c.h for c in ctxs])
    rc = lib.clc_essential_acransac_batch(hs, jobs, n)
    if rc != CLC_OK:
        raise CLCError(rc, lib.clc_status_string(rc).decode())
    return [_two_view_result(j, *o) for j, o in zip(jobs, outs)]


def two_view_acransac_batch(ctxs, model, problems, max_iteration=256, precision=float("inf")):
    """clc_two_view_acransac_batch: model 'E' | 'F' | 'H'; problems = [(x1, x2, K1, K2, (w, h), seed), ...] (K1 / K2 read for 'E' only), one
    Context per problem; -> list of result dicts, "E" = the model matrix (E, F or H in pixels)."""
    lib = load_library()
    n = len(problems)
    jobs = (TwoViewJob * n)()
    keep, outs = [], []
    for j, (x1, x2, K1, K2, wh, seed) in zip(jobs, problems):
        outs.append(_two_view_fill(j, keep, x1, x2, np.eye(3) if K1 is None else K1, np.eye(3) if K2 is None else K2, wh, max_iteration, seed, precision))
    hs = (C.c_void_p * n)(*[c.h for c in ctxs])
    rc = lib.clc_two_view_acransac_batch(hs, ord(model), jobs, n)
    if rc != CLC_OK:
        raise CLCError(rc, lib.clc_status_string(rc).decode())
    return [_two_view_result(j, *o) for j, o in zip(jobs, outs)]


def inter_pose_batch(ctxs, problems, map_X, max_iteration=256, huber_a=16.0):
    """clc_inter_pose_batch: problems = [dict(x1, x2, K, wh, seed, Rt_source, and map_index (the shortcut) OR d_first_desc, first_feature,
    d_map_desc [, match_threshold] (the reference's chain: device addresses of the lower camera's descriptor block and of the global
    map's descriptors, the correspondences' rows in the former)), ...] (x1 = source frame's features, x2 = the destination's), one
    Context per problem, map_X the global map's points (M x 3); -> list of dicts (two-view result + Rt, cov, rmse, scale, n_front,
    n_common, n_map_matches, stage)."""
    lib = load_library()
    n = len(problems)
    jobs = (InterPoseJob * n)()
    map_X = np.ascontiguousarray(map_X, dtype=np.float64).reshape(-1, 3)
    keep, outs = [map_X], []
    for j, p in zip(jobs, problems):
        outs.append(_two_view_fill(j.tv, keep, p["x1"], p["x2"], p["K"], p["K"], p["wh"], max_iteration, p["seed"], float("inf")))
        rs = np.ascontiguousarray(p["Rt_source"], dtype=np.float64).reshape(12)
        keep.append(rs)
        j.map_X, j.map_n, j.Rt_source, j.huber_a = map_X.ctypes.data, int(map_X.shape[0]), rs.ctypes.data, float(huber_a)
        if p.get("d_first_desc") is not None:
            ff = np.ascontiguousarray(p["first_feature"], dtype=np.int32)
            keep.append(ff)
            j.d_first_desc, j.first_feature, j.d_map_desc = int(p["d_first_desc"]), ff.ctypes.data, int(p["d_map_desc"])
            j.match_threshold = int(p.get("match_threshold", 60))
        elif p.get("map_index") is not None:
            mi = np.ascontiguousarray(p["map_index"], dtype=np.int32)
            keep.append(mi)
            j.map_index = mi.ctypes.data
    hs = (C.c_void_p * n)(*[c.h for c in ctxs])
    rc = lib.clc_inter_pose_batch(hs, jobs, n)
    if rc != CLC_OK:
        raise CLCError(rc, lib.clc_status_string(rc).decode())
    res = []
    for j, o in zip(jobs, outs):
        d = _two_view_result(j.tv, *o)
        d.update(Rt=np.array(j.Rt).reshape(3, 4), cov=np.array(j.cov).reshape(6, 6), rmse=j.rmse, scale=j.scale, n_front=j.n_front,
                 n_common=j.n_common, n_refined=j.n_refined, stage=j.stage, n_map_matches=j.n_map_matches)
        res.append(d)
    return res


def mc_plan(counts, world, rank, grain):
    """clc_mc_plan: the shares of `rank` as a list of (first, second, q_begin, nq, out_offset)."""
    lib = load_library()
    cnt = (C.c_int * len(counts))(*[int(c) for c in counts])
    cap = len(counts) * len(counts) + 2
    out = (McShare * cap)()
    n = C.c_int()
    rc = lib.clc_mc_plan(cnt, len(counts), int(world), int(rank), int(grain), out, cap, C.byref(n))
    if rc != CLC_OK:
        raise CLCError(rc, lib.clc_status_string(rc).decode())
    return [(s.first, s.second, s.q_begin, s.nq, s.out_offset) for s in out[:n.value]]


class MultiCam:
    """clc_mc handle: one rank of the multi-camera exchange + match step (world = 1: no RCCL needed)."""

    def __init__(self, ctx, world=1, rank=0, maxkp=10000, unique_id=None):
        self.lib, self.ctx = load_library(), ctx
        h = C.c_void_p()
        idbuf = (C.c_uint8 * 128)(*unique_id) if unique_id is not None else None
        rc = self.lib.clc_mc_create(ctx.h, idbuf, world, rank, maxkp, C.byref(h))
        if rc != CLC_OK:
            raise CLCError(rc, "clc_mc_create: " + self.lib.clc_status_string(rc).decode())
        self.h, self.world, self.maxkp = h, world, maxkp

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        rc = lib.clc_mc_unique_id(buf)
        if rc != CLC_OK:
            raise CLCError(rc, "clc_mc_unique_id: " + lib.clc_status_string(rc).decode())
        return bytes(buf)

    def _chk(self, rc):
        if rc != CLC_OK:
            raise CLCError(rc, "%s: %s" % (self.lib.clc_status_string(rc).decode(), self.lib.clc_mc_last_error_string(self.h).decode()))

    def arena(self):
        p = C.c_void_p()
        self._chk(self.lib.clc_mc_arena(self.h, C.byref(p), None, None))
        return p.value

    def gather_dev(self, d_my_desc, my_count, mode=0, stream=None):
        cnt = (C.c_int * self.world)()
        self._chk(self.lib.clc_mc_gather_dev(self.h, d_my_desc, int(my_count), int(mode), cnt, stream))
        return [int(c) for c in cnt]

    def virtual_put(self, other_rank, d_desc, count, stream=None):
        self._chk(self.lib.clc_mc_virtual_put(self.h, int(other_rank), d_desc, int(count), stream))

    def open_peers(self, stream=None):
        """Collective: map the peers' arenas now (the first peer-copy exchange would do it otherwise)."""
        self._chk(self.lib.clc_mc_open_peers(self.h, stream))

    def match_dev(self, threshold, d_match, capacity, stream=None):
        cap = self.world * self.world + 2
        sh = (McShare * cap)()
        n = C.c_int()
        self._chk(self.lib.clc_mc_match_dev(self.h, int(threshold), d_match, int(capacity), sh, cap, C.byref(n), stream))
        return [(s.first, s.second, s.q_begin, s.nq, s.out_offset) for s in sh[:n.value]]

    def gather_enqueue_dev(self, d_my_desc, my_count, mode=0, stream=None, d_my_count=None):
        """Enqueue-only exchange (no host synchronisation); my_count from the host, or d_my_count: device address of an int32."""
        self._chk(self.lib.clc_mc_gather_enqueue_dev(self.h, d_my_desc, int(my_count), d_my_count, int(mode), stream))

    def match_enqueue_dev(self, threshold, d_match, capacity, stream=None):
        """Capacity-planned shares, counts read on the device: pairs with clc_mc_gather_enqueue_dev."""
        cap = self.world * self.world + 2
        sh = (McShare * cap)()
        n = C.c_int()
        self._chk(self.lib.clc_mc_match_enqueue_dev(self.h, int(threshold), d_match, int(capacity), sh, cap, C.byref(n), stream))
        return [(s.first, s.second, s.q_begin, s.nq, s.out_offset) for s in sh[:n.value]]

    def set_overlap(self, on=True):
        """clc_mc_set_overlap: exchange (+ the caller's describe) on one stream, the sweep on another; before the first exchange."""
        self._chk(self.lib.clc_mc_set_overlap(self.h, 1 if on else 0))

    def comm_info(self):
        """(ncclCommCount, ncclCommUserRank) of the communicator behind the handle; (0, -1): none."""
        n, r = C.c_int(), C.c_int()
        self._chk(self.lib.clc_mc_comm_info(self.h, C.byref(n), C.byref(r)))
        return int(n.value), int(r.value)

    def counts(self, stream=None):
        cnt = (C.c_int * self.world)()
        self._chk(self.lib.clc_mc_counts(self.h, cnt, stream))
        return [int(c) for c in cnt]

    def close(self):
        if getattr(self, "h", None):
            self.lib.clc_mc_destroy(self.h)
            self.h = None


def pnp_localize_batch(ctxs, problems, max_iteration=256, seeds=None, precision=float("inf"), refine=True, huber_a=16.0):
    """clc_pnp_localize_ac_batch: problems = [(X, x, K), ...], one context per problem (all on one device); the a-contrario solves run
    interleaved on the GPU, driven by one host thread.  Returns a list of dicts like Context.pnp_acransac's."""
    lib = load_library()
    n = len(problems)
    assert len(ctxs) == n
    jobs = (PoseJob * n)()
    keep = []
    for i, (X, x, K) in enumerate(problems):
        X = np.ascontiguousarray(X, dtype=np.float64); x = np.ascontiguousarray(x, dtype=np.float64)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        N = X.shape[0]
        Rt, cov = np.zeros(12), np.zeros(36)
        mask, inl = np.zeros(max(N, 1), dtype=np.uint8), np.zeros(max(N, 1), dtype=np.int32)
        keep.append((X, x, K, Rt, cov, mask, inl))
        j = jobs[i]
        j.X, j.x, j.K, j.n = X.ctypes.data, x.ctypes.data, K.ctypes.data, N
        j.max_iteration, j.seed, j.precision = int(max_iteration), int(seeds[i] if seeds is not None else 1), float(precision)
        j.refine, j.huber_a = (1 if refine else 0), float(huber_a)
        j.Rt, j.cov, j.inlier_mask, j.inliers = Rt.ctypes.data, cov.ctypes.data, mask.ctypes.data, inl.ctypes.data
    hs = (C.c_void_p * n)(*[c.h for c in ctxs])
    rc = lib.clc_pnp_localize_ac_batch(hs, jobs, n)
    if rc != CLC_OK:
        raise CLCError(rc, "clc_pnp_localize_ac_batch: " + lib.clc_status_string(rc).decode())
    out = []
    for i in range(n):
        X, x, K, Rt, cov, mask, inl = keep[i]
        j = jobs[i]
        found = j.n_inliers > 0
        out.append(dict(Rt=Rt.reshape(3, 4) if found else None, cov=cov.reshape(6, 6) if (found and refine) else None,
                        mask=mask[:X.shape[0]].astype(bool), inliers=inl[:j.n_inliers].copy(), error_max=j.error_max, rmse=j.rmse,
                        iterations=j.iterations))
    return out


def desc_handle_live(handle):
    """1 while the publication the handle came from still stands (same host address, count and generation)."""
    return bool(load_library().clc_desc_handle_live(C.byref(handle)))


def desc_cache_stats():
    """(hits, misses) of the descriptor cache behind the host-pointer match entry points."""
    lib = load_library()
    h, m = C.c_ulonglong(), C.c_ulonglong()
    lib.clc_desc_cache_stats(C.byref(h), C.byref(m))
    return int(h.value), int(m.value)


def keypoints_to_features(kps):
    lib = load_library()
    kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
    out = np.zeros((len(kps), 4), dtype=np.float32)
    rc = lib.clc_keypoints_to_features(_p(kps), len(kps), _p(out))
    if rc != CLC_OK:
        raise CLCError(rc, lib.clc_status_string(rc).decode())
    return out


class Context:
    """One clc_ctx: device buffers + stream on one GPU (one per host thread / per rank)."""

    def __init__(self, device=0, width=640, height=480, maxkp=10000, scale_factor=1.2, scale_levels=8,
                 fast_thresh=40, match_thresh=60, detector=True, matcher=True):
        self.lib = load_library()
        self.dopts = DetectorOptions(scale_factor, scale_levels, width, height, maxkp, fast_thresh)
        self.mopts = MatcherOptions(0.8, match_thresh, maxkp)
        h = C.c_void_p()
        rc = self.lib.clc_ctx_create(device, C.byref(self.dopts) if detector else None,
                                     C.byref(self.mopts) if matcher else None, C.byref(h))
        if rc != CLC_OK:
            raise CLCError(rc, "clc_ctx_create: " + self.lib.clc_status_string(rc).decode())
        self.h = h
        self.device = device

    def _chk(self, rc):
        if rc != CLC_OK:
            raise CLCError(rc, "%s: %s" % (self.lib.clc_status_string(rc).decode(),
                                           self.lib.clc_last_error_string(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.clc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self.lib.clc_sync(self.h))

    @property
    def stream(self):
        return self.lib.clc_stream(self.h)

    # -- per-kernel event timing
    def profile_enable(self, on=True, only=None):
        """on=True brackets every kernel with HIP events; only=[kernel names] restricts it."""
        v = 1 if on else 0
        if on and only:
            v = 0
            for name in only:
                v |= 1 << (KERNELS.index(name) + 1)
        self._chk(self.lib.clc_profile_enable(self.h, v))

    def profile_reset(self):
        self._chk(self.lib.clc_profile_reset(self.h))

    def profile_read(self):
        """{kernel name: (total device ms, launches)} since the last reset (synchronises)."""
        out = {}
        for k, name in enumerate(KERNELS):
            ms, cnt = C.c_double(), C.c_int()
            self._chk(self.lib.clc_profile_read(self.h, k, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    # -- pyramid
    def pyramid_build(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self._chk(self.lib.clc_pyramid_build(self.h, _p(img), img.shape[1], img.shape[0]))

    def pyramid_build_dev(self, d_ptr, width, height, pitch, stream=None):
        self._chk(self.lib.clc_pyramid_build_dev(self.h, d_ptr, width, height, pitch, stream))

    def pyramid_level(self, level):
        w, h, p, d = C.c_uint32(), C.c_uint32(), C.c_size_t(), C.c_void_p()
        self._chk(self.lib.clc_pyramid_level(self.h, level, C.byref(w), C.byref(h), C.byref(p), C.byref(d)))
        return w.value, h.value, p.value, d.value

    def pyramid_download(self, level):
        w, h, _, _ = self.pyramid_level(level)
        out = np.zeros((h, w), dtype=np.uint8)
        self._chk(self.lib.clc_pyramid_download(self.h, level, _p(out)))
        return out

    # -- detect (GPU FAST-9 + NMS + orientation)
    def detect(self, capacity=None):
        """Keypoints of the current pyramid in the reference's order; returns (kps, n_found)."""
        cap = int(capacity if capacity is not None else self.dopts.maxkp)
        kps = np.zeros(cap, dtype=KP_DTYPE)
        n, found = C.c_int(), C.c_int()
        self._chk(self.lib.clc_detect(self.h, _p(kps), cap, C.byref(n), C.byref(found)))
        return kps[:n.value], found.value

    def detect_dev(self, stream=None):
        self._chk(self.lib.clc_detect_dev(self.h, stream))

    def detect_buffers(self):
        k, c, d = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.lib.clc_detect_buffers(self.h, C.byref(k), C.byref(c), C.byref(d)))
        return k.value, c.value, d.value

    def describe_detected_dev(self, d_desc=None, stream=None):
        self._chk(self.lib.clc_describe_detected_dev(self.h, d_desc, stream))

    def detect_and_describe(self, img, capacity=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        cap = int(capacity if capacity is not None else self.dopts.maxkp)
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 64), dtype=np.uint8)
        n, found = C.c_int(), C.c_int()
        self._chk(self.lib.clc_detect_and_describe(self.h, _p(img), img.shape[1], img.shape[0], _p(kps), _p(desc), cap,
                                                   C.byref(n), C.byref(found)))
        return kps[:n.value], desc[:n.value], found.value

    def detect_and_describe_published(self, img):
        """The policy classes' front end: clc_detect_and_describe_view (one enqueue sequence, one synchronisation, results in the
        context's pinned block) + clc_detect_store_descriptors (the frame's one copy into the caller's block, published on the way).
        Returns (keypoints, descriptors, found, DescHandle)."""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        pk, pd, n, found = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        self._chk(self.lib.clc_detect_and_describe_view(self.h, _p(img), img.shape[1], img.shape[0], C.byref(pk), C.byref(pd), C.byref(n), C.byref(found)))
        kps = np.zeros(n.value, dtype=KP_DTYPE)
        if n.value:
            C.memmove(_p(kps), pk, n.value * 20)
        desc = np.zeros((n.value, 64), dtype=np.uint8)
        handle = DescHandle()
        self._chk(self.lib.clc_detect_store_descriptors(self.h, _p(desc) if n.value else None, n.value, C.byref(handle)))
        return kps, desc, found.value, handle

    def detect_batch_dev(self, d_imgs, width, height, pitch, d_kps, d_counts, d_desc=None, stream=None):
        """Frames of several cameras (lists of device pointers): one pyramid launch, two detector launches and (with d_desc) one
        CLATCH launch; d_kps[b] holds maxkp keypoints, d_counts[b] a uint32 pair {written, found}, d_desc[b] maxkp x 64 B."""
        n = len(d_imgs)
        imgs = (C.c_void_p * n)(*d_imgs)
        kps = (C.c_void_p * n)(*d_kps)
        cnt = (C.c_void_p * n)(*d_counts)
        out = (C.c_void_p * n)(*d_desc) if d_desc is not None else None
        self._chk(self.lib.clc_detect_batch_dev(self.h, n, imgs, width, height, pitch, kps, cnt, out, stream))

    def describe_match_pair_dev(self, d_imgs, width, height, pitch, d_kps, counts, d_desc, threshold, d_match, stream=None):
        """clc_describe_match_pair_dev: pyramid + CLATCH of the pair's two cameras and the sweep camera 0 -> camera 1 as one enqueue."""
        imgs = (C.c_void_p * 2)(*d_imgs)
        kps = (C.c_void_p * 2)(*d_kps)
        cnt = (C.c_int * 2)(*[int(c) for c in counts])
        out = (C.c_void_p * 2)(*d_desc)
        self._chk(self.lib.clc_describe_match_pair_dev(self.h, imgs, width, height, pitch, kps, cnt, out, int(threshold), d_match, stream))

    def desc_cache_mode(self, mode):
        """clc_desc_cache_mode: "off" | "verify" (default: sweep at once, whole-block fold behind it, edited blocks uploaded) | "trust"."""
        self._chk(self.lib.clc_desc_cache_mode(self.h, {"off": DESC_CACHE_OFF, "verify": DESC_CACHE_VERIFY, "trust": DESC_CACHE_TRUST}[mode]))

    def desc_cache_publish(self, h_desc, d_src=None):
        """The rows of numpy block h_desc (n x 64) are on this device at d_src (None: the context's own descriptor array, what
        detect_and_describe filled): later host-pointer match calls given this very block skip its upload."""
        assert h_desc.dtype == np.uint8 and h_desc.flags["C_CONTIGUOUS"]
        handle = DescHandle()
        self._chk(self.lib.clc_desc_cache_publish(self.h, d_src, _p(h_desc), int(h_desc.shape[0]), C.byref(handle)))
        return handle

    # -- describe
    def describe(self, kps):
        kps = np.ascontiguousarray(kps, dtype=KP_DTYPE)
        desc = np.zeros((len(kps), 64), dtype=np.uint8)
        self._chk(self.lib.clc_describe(self.h, _p(kps), len(kps), _p(desc)))
        return desc

    def describe_dev(self, d_kps, n, d_desc, stream=None):
        self._chk(self.lib.clc_describe_dev(self.h, d_kps, n, d_desc, stream))

    def describe_batch_dev(self, d_imgs, width, height, pitch, d_kps, counts, d_desc, stream=None):
        """Frames of several cameras (lists of device pointers / counts) in one pyramid + one CLATCH launch."""
        n = len(d_imgs)
        imgs = (C.c_void_p * n)(*d_imgs)
        kps = (C.c_void_p * n)(*d_kps)
        cnt = (C.c_int * n)(*counts)
        out = (C.c_void_p * n)(*d_desc)
        self._chk(self.lib.clc_describe_batch_dev(self.h, n, imgs, width, height, pitch, kps, cnt, out, stream))

    # -- match
    def set_k2nn_formulation(self, name):
        """"matrix" (FP4 matrix pipe, default) or "popcount" (xor + popcount on the vector ALU): same results."""
        self._chk(self.lib.clc_k2nn_set_formulation(self.h, {"matrix": 0, "popcount": 1}[name]))

    @property
    def k2nn_queries_per_block(self):
        return int(self.lib.clc_k2nn_queries_per_block(self.h))

    def k2nn_device_info(self):
        """clc_k2nn_device_info: what the sweep planner knows about the device and where its unequal shares come from (dict)."""
        info = (C.c_int32 * 8)()
        us = (C.c_float * 4)()
        self._chk(self.lib.clc_k2nn_device_info(self.h, info, us))
        d = dict(zip(("xcds", "cus", "default_target_blocks", "bias_a", "bias_b", "bias_source", "kernel_xcds", "target_blocks_env"), [int(v) for v in info]))
        d["bias_source"] = ("default", "CLC_K2NN_BIAS", "probe")[d["bias_source"]]
        d["probe_us"] = dict(zip(("295:264", "311:256", "326:249", "326:233"), [float(v) for v in us]))
        return d

    def k2nn_plan_query(self, nq, nt):
        """clc_k2nn_plan_query: how one nq x nt pair would be cut into sweep workgroups (dict)."""
        info = (C.c_int32 * 8)()
        self._chk(self.lib.clc_k2nn_plan_query(self.h, int(nq), int(nt), info))
        return dict(zip(("qblocks", "splits", "t_per_split", "atomic_merge", "bias_a_tiles", "bias_b_tiles", "queries_per_block", "target_blocks"),
                        [int(v) for v in info]))

    def k2nn_clock_check(self, d_q, nq, d_t, nt, d_match, stream=None):
        """In-kernel shader clock of one (diagnostic, stamped) sweep: (median GHz, min, max, workgroups)."""
        med, lo, hi, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        self._chk(self.lib.clc_k2nn_clock_check(self.h, d_q, nq, d_t, nt, d_match, stream, C.byref(med), C.byref(lo), C.byref(hi), C.byref(n)))
        return med.value, lo.value, hi.value, n.value

    def match_2nn(self, Q, T, threshold=40, want_dist=False):
        Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 64)
        T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 64)
        m = np.full(Q.shape[0], -2, dtype=np.int32)
        b = np.zeros(Q.shape[0], dtype=np.uint16) if want_dist else None
        s = np.zeros(Q.shape[0], dtype=np.uint16) if want_dist else None
        self._chk(self.lib.clc_match_2nn(self.h, _p(Q), Q.shape[0], _p(T), T.shape[0], int(threshold), _p(m), _p(b), _p(s)))
        return (m, b, s) if want_dist else m

    def match_2nn_dev(self, d_q, nq, d_t, nt, threshold, d_match, stream=None):
        self._chk(self.lib.clc_match_2nn_dev(self.h, d_q, nq, d_t, nt, int(threshold), d_match, stream))

    def match_jobs_dev(self, d_desc_base, jobs, d_match, stream=None):
        arr = (MatchJob * len(jobs))(*[MatchJob(*j) for j in jobs])
        self._chk(self.lib.clc_match_jobs_dev(self.h, d_desc_base, arr, len(jobs), d_match, stream))

    def match_jobs_counted_dev(self, d_desc_base, jobs, d_cnt_q, d_cnt_t, q_row0, d_match, stream=None):
        """clc_match_jobs_counted_dev: jobs planned on capacities, the row counts of both sets read on the device (lists of device
        addresses of int32 counts, one per job); planned query rows past the count are answered -1."""
        n = len(jobs)
        arr = (MatchJob * n)(*[MatchJob(*j) for j in jobs])
        cq = (C.c_void_p * n)(*d_cnt_q)
        ct = (C.c_void_p * n)(*d_cnt_t)
        r0 = (C.c_uint32 * n)(*q_row0)
        self._chk(self.lib.clc_match_jobs_counted_dev(self.h, d_desc_base, arr, n, cq, ct, r0, d_match, stream))

    def match_pairs(self, descs, pairs, threshold=40):
        """All listed (first, second) pairs over per-camera descriptor arrays; returns a list of int32 arrays."""
        descs = [np.ascontiguousarray(d, dtype=np.uint8).reshape(-1, 64) for d in descs]
        counts = (C.c_int * len(descs))(*[d.shape[0] for d in descs])
        dptr = (C.c_void_p * len(descs))(*[d.ctypes.data for d in descs])
        flat = (C.c_int * (2 * len(pairs)))(*[v for p in pairs for v in p])
        outs = [np.full(descs[a].shape[0], -2, dtype=np.int32) for a, _ in pairs]
        optr = (C.c_void_p * len(pairs))(*[o.ctypes.data for o in outs])
        self._chk(self.lib.clc_match_pairs(self.h, dptr, counts, len(descs), flat, len(pairs), int(threshold), optr))
        return outs

    def set_map(self, desc):
        desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(-1, 64)
        self._chk(self.lib.clc_set_map(self.h, _p(desc), desc.shape[0]))

    def match_map_dev(self, d_q, nq, threshold, d_match, stream=None):
        self._chk(self.lib.clc_match_map_dev(self.h, d_q, nq, int(threshold), d_match, stream))

    def match_map(self, Q, threshold=60):
        Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 64)
        m = np.full(Q.shape[0], -2, dtype=np.int32)
        self._chk(self.lib.clc_match_map(self.h, _p(Q), Q.shape[0], int(threshold), _p(m)))
        return m

    # -- pnp
    def pnp_residuals(self, Rt, X, x, K):
        Rt = np.ascontiguousarray(Rt, dtype=np.float64).reshape(-1, 12)
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        err = np.zeros((Rt.shape[0], X.shape[0]), dtype=np.float64)
        self._chk(self.lib.clc_pnp_residuals(self.h, _p(Rt), Rt.shape[0], _p(X), _p(x), X.shape[0], _p(K), _p(err)))
        return err

    def pnp_ransac(self, X, x, K, samples=None, n_samples=256, seed=1, thr2=16.0):
        """Robust pose: (Rt (3,4) or None, inlier mask, cost)."""
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        if samples is not None:
            samples = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 3)
            n_samples = samples.shape[0]
        Rt = np.zeros(12, dtype=np.float64)
        mask = np.zeros(X.shape[0], dtype=np.uint8)
        n, cost = C.c_int(), C.c_double()
        self._chk(self.lib.clc_pnp_ransac(self.h, _p(X), _p(x), X.shape[0], _p(K), _p(samples), int(n_samples), int(seed),
                                          float(thr2), _p(Rt), _p(mask), C.byref(n), C.byref(cost)))
        return (Rt.reshape(3, 4) if n.value > 0 else None), mask.astype(bool), cost.value

    def essential_ransac(self, x1, x2, K1, K2, samples=None, n_samples=256, seed=1, thr2=4.0):
        """Five-point RANSAC: (E (3,3) or None, F (3,3), inlier mask)."""
        x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
        K1 = np.ascontiguousarray(K1, dtype=np.float64).reshape(9); K2 = np.ascontiguousarray(K2, dtype=np.float64).reshape(9)
        if samples is not None:
            samples = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 5)
            n_samples = samples.shape[0]
        E = np.zeros(9); F = np.zeros(9)
        mask = np.zeros(x1.shape[0], dtype=np.uint8)
        n = C.c_int()
        self._chk(self.lib.clc_essential_ransac(self.h, _p(x1), _p(x2), x1.shape[0], _p(K1), _p(K2), _p(samples), int(n_samples),
                                                int(seed), float(thr2), _p(E), _p(F), _p(mask), C.byref(n)))
        return (E.reshape(3, 3) if n.value > 0 else None), F.reshape(3, 3), mask.astype(bool)

    def essential_fivepoint(self, x1, x2, K1, K2, samples):
        x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
        K1 = np.ascontiguousarray(K1, dtype=np.float64).reshape(9); K2 = np.ascontiguousarray(K2, dtype=np.float64).reshape(9)
        samples = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 5)
        out = np.zeros((samples.shape[0], 10, 9), dtype=np.float64)
        self._chk(self.lib.clc_essential_fivepoint(self.h, _p(x1), _p(x2), x1.shape[0], _p(K1), _p(K2), _p(samples),
                                                   samples.shape[0], _p(out)))
        return out

    def epipolar_residuals(self, F, x1, x2):
        F = np.ascontiguousarray(F, dtype=np.float64).reshape(-1, 9)
        x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
        err = np.zeros((F.shape[0], x1.shape[0]), dtype=np.float64)
        self._chk(self.lib.clc_epipolar_residuals(self.h, _p(F), F.shape[0], _p(x1), _p(x2), x1.shape[0], _p(err)))
        return err

    def epipolar_score(self, F, x1, x2, thr2):
        F = np.ascontiguousarray(F, dtype=np.float64).reshape(-1, 9)
        x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
        cnt = np.zeros(F.shape[0], dtype=np.int32); cost = np.zeros(F.shape[0], dtype=np.float64)
        self._chk(self.lib.clc_epipolar_score(self.h, _p(F), F.shape[0], _p(x1), _p(x2), x1.shape[0], float(thr2), _p(cnt), _p(cost)))
        return cnt, cost

    def pnp_localize(self, X, x, K, samples=None, n_samples=256, seed=1, thr2=16.0, huber_a=16.0):
        """ransac + refine in one submission: (Rt (3,4) or None, cov (6,6), inlier mask, rmse)."""
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        if samples is not None:
            samples = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 3)
            n_samples = samples.shape[0]
        Rt = np.zeros(12); cov = np.zeros(36)
        mask = np.zeros(X.shape[0], dtype=np.uint8)
        n, rmse = C.c_int(), C.c_double()
        self._chk(self.lib.clc_pnp_localize(self.h, _p(X), _p(x), X.shape[0], _p(K), _p(samples), int(n_samples), int(seed),
                                            float(thr2), float(huber_a), _p(Rt), _p(cov), _p(mask), C.byref(n), C.byref(rmse)))
        return (Rt.reshape(3, 4) if n.value > 0 else None), cov.reshape(6, 6), mask.astype(bool), rmse.value

    def pnp_acransac(self, X, x, K, max_iteration=256, seed=1, precision=float("inf"), refine=False, huber_a=16.0):
        """A-contrario RANSAC pose (what the reference runs).  Returns a dict: Rt (3,4) or None, mask, inliers (ascending
        residual order), error_max (pixels), min_nfa, iterations [, cov, rmse with refine=True]."""
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        n = X.shape[0]
        Rt = np.zeros(12); mask = np.zeros(max(n, 1), dtype=np.uint8); inl = np.zeros(max(n, 1), dtype=np.int32)
        ni, its = C.c_int(), C.c_int()
        emax, nfa, rmse = C.c_double(), C.c_double(), C.c_double()
        out = {}
        if refine:
            cov = np.zeros(36)
            self._chk(self.lib.clc_pnp_localize_ac(self.h, _p(X), _p(x), n, _p(K), int(max_iteration), int(seed), float(precision),
                                                   float(huber_a), _p(Rt), _p(cov), _p(mask), _p(inl), C.byref(ni), C.byref(emax),
                                                   C.byref(rmse)))
            out.update(cov=cov.reshape(6, 6), rmse=rmse.value)
        else:
            self._chk(self.lib.clc_pnp_acransac(self.h, _p(X), _p(x), n, _p(K), int(max_iteration), int(seed), float(precision),
                                                _p(Rt), _p(mask), _p(inl), C.byref(ni), C.byref(emax), C.byref(nfa), C.byref(its)))
            out.update(min_nfa=nfa.value, iterations=its.value)
        out.update(Rt=Rt.reshape(3, 4) if ni.value > 0 else None, mask=mask[:n].astype(bool), inliers=inl[:ni.value].copy(),
                   error_max=emax.value)
        return out

    def essential_acransac(self, x1, x2, K1, K2, img_wh, max_iteration=256, seed=1, precision=float("inf")):
        """A-contrario five-point RANSAC (RobustMatcher::filterEssential).  Returns a dict: E, F (3,3) or None, mask, inliers, ..."""
        x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
        K1 = np.ascontiguousarray(K1, dtype=np.float64).reshape(9); K2 = np.ascontiguousarray(K2, dtype=np.float64).reshape(9)
        n = x1.shape[0]
        E = np.zeros(9); F = np.zeros(9)
        mask = np.zeros(max(n, 1), dtype=np.uint8); inl = np.zeros(max(n, 1), dtype=np.int32)
        ni, its = C.c_int(), C.c_int()
        emax, nfa = C.c_double(), C.c_double()
        self._chk(self.lib.clc_essential_acransac(self.h, _p(x1), _p(x2), n, _p(K1), _p(K2), int(img_wh[0]), int(img_wh[1]),
                                                  int(max_iteration), int(seed), float(precision), _p(E), _p(F), _p(mask), _p(inl),
                                                  C.byref(ni), C.byref(emax), C.byref(nfa), C.byref(its)))
        ok = ni.value > 0
        return dict(E=E.reshape(3, 3) if ok else None, F=F.reshape(3, 3) if ok else None, mask=mask[:n].astype(bool),
                    inliers=inl[:ni.value].copy(), error_max=emax.value, min_nfa=nfa.value, iterations=its.value)

    def two_view_acransac(self, model, x1, x2, img_wh, K1=None, K2=None, max_iteration=256, seed=1, precision=float("inf")):
        """clc_two_view_acransac: the a-contrario filter under model 'E' (five points; K1, K2 needed), 'F' (seven points) or 'H' (four points).
        Returns a dict: M (3,3: E, F or H in pixels) or None, F (3,3; zeros for 'H') or None, mask, inliers, error_max, min_nfa, iterations."""
        x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
        K1 = None if K1 is None else np.ascontiguousarray(K1, dtype=np.float64).reshape(9)
        K2 = None if K2 is None else np.ascontiguousarray(K2, dtype=np.float64).reshape(9)
        n = x1.shape[0]
        M = np.zeros(9); F = np.zeros(9)
        mask = np.zeros(max(n, 1), dtype=np.uint8); inl = np.zeros(max(n, 1), dtype=np.int32)
        ni, its = C.c_int(), C.c_int()
        emax, nfa = C.c_double(), C.c_double()
        self._chk(self.lib.clc_two_view_acransac(self.h, ord(model), _p(x1), _p(x2), n, _p(K1), _p(K2), int(img_wh[0]), int(img_wh[1]),
                                                 int(max_iteration), int(seed), float(precision), _p(M), _p(F), _p(mask), _p(inl),
                                                 C.byref(ni), C.byref(emax), C.byref(nfa), C.byref(its)))
        ok = ni.value > 0
        return dict(M=M.reshape(3, 3) if ok else None, F=F.reshape(3, 3) if ok else None, mask=mask[:n].astype(bool),
                    inliers=inl[:ni.value].copy(), error_max=emax.value, min_nfa=nfa.value, iterations=its.value)

    def two_view_minimal(self, model, x1, x2, img_wh, samples):
        """clc_two_view_minimal: the seven-point ('F') / four-point ('H') models of the given samples (S x 7 | 4 indices), in the
        coordinates conditioned by the image size: array (S, 3 | 1, 9), NaN rows where a sample has fewer real roots."""
        x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
        m, M = (7, 3) if model == "F" else (4, 1)
        smp = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, m)
        out = np.zeros((smp.shape[0], M, 9))
        self._chk(self.lib.clc_two_view_minimal(self.h, ord(model), _p(x1), _p(x2), x1.shape[0], int(img_wh[0]), int(img_wh[1]), _p(smp),
                                                smp.shape[0], _p(out)))
        return out

    def pnp_refine(self, X, x, K, Rt0, mask=None, huber_a=16.0, max_iter=50):
        """LM refinement of one pose: returns (Rt (3,4), cov (6,6), rmse, iterations)."""
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        Rt0 = np.ascontiguousarray(Rt0, dtype=np.float64).reshape(12)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        Rt = np.zeros(12); cov = np.zeros(36)
        rmse, it = C.c_double(), C.c_int()
        self._chk(self.lib.clc_pnp_refine(self.h, _p(X), _p(x), X.shape[0], _p(K), _p(m), _p(Rt0), float(huber_a), int(max_iter),
                                          _p(Rt), _p(cov), C.byref(rmse), C.byref(it)))
        return Rt.reshape(3, 4), cov.reshape(6, 6), rmse.value, it.value

    def pnp_p3p(self, X, x, K, samples):
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        samples = np.ascontiguousarray(samples, dtype=np.int32).reshape(-1, 3)
        out = np.zeros((samples.shape[0], 4, 12), dtype=np.float64)
        self._chk(self.lib.clc_pnp_p3p(self.h, _p(X), _p(x), X.shape[0], _p(K), _p(samples), samples.shape[0], _p(out)))
        return out

    def pnp_score(self, Rt, X, x, K, thr2):
        Rt = np.ascontiguousarray(Rt, dtype=np.float64).reshape(-1, 12)
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        cnt = np.zeros(Rt.shape[0], dtype=np.int32)
        cost = np.zeros(Rt.shape[0], dtype=np.float64)
        self._chk(self.lib.clc_pnp_score(self.h, _p(Rt), Rt.shape[0], _p(X), _p(x), X.shape[0], _p(K), float(thr2), _p(cnt), _p(cost)))
        return cnt, cost
