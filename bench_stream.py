#!/usr/bin/env python3
"""bench_stream.py -- BASELINE config[4] shaped run: cameras streaming synthetic video through the whole
loop describe -> match against the map -> robust pose + covariance -> covariance-intersection fusion.

NOT the driver's benchmark (that is bench.py); this is the scenario harness for the streaming
configuration.  One camera per rank (python -m torch.distributed.run --nproc-per-node N bench_stream.py
--gpus N); with N = 1 it runs `--cams` cameras round-robin on the one GPU.

Two scenes:
  --scene plane (default): every frame is RENDERED from the camera's pose over a textured plane (tests/synth.py
     render_plane), the map is the CLATCH output of a reference view with its 3-D points, and stage 2 matches the
     descriptors stage 1 has just computed -- they never leave the device; stage 3 is the a-contrario P3P of the
     reference (Localizer.hpp:82-93) + refinement.  Nothing is mocked; the recovered position is checked against
     the pose the frame was rendered from.
  --scene synthetic: the round-1 harness (frames without geometry; observed descriptors synthesised from the map).

Per frame and camera (reference include/coloc/coloc.hpp:201-272 intraPoseEstimator + :362-389 fusion):
  1. front end on the frame image, device-resident: pyramid -> FAST-9/NMS/orientation -> CLATCH
     (--scene synthetic: the frames carry no geometry, so these descriptors are not matched);
  2. map tracking: the frame's observed descriptors (map descriptors with bit noise + distractors; device-resident
     like the CLATCH output they stand in for) against the map on the GPU (clc_match_map_dev), threshold MatcherOptions.thresh = 60 -> 2D-3D correspondences (GPUMatcher.hpp:174-178,252-271);
  3. clc_pnp_localize: 256 P3P samples -> scored hypotheses -> LM refinement + 6x6 covariance;
  4. inter-camera step (plane scene; reference include/coloc/coloc.hpp:274-392 interPoseEstimator(source, dest)): the two
     frames' CLATCH outputs matched on the device (computeMatchesPair, :287), a-contrario five-point filter + relative pose from
     E with the chirality vote (filterMatchesPair, :296), the pair's temporary map triangulated and brought to the global map's
     scale through the features both maps hold (map matches, threshold 60, :323-336), the destination pose refined against it
     (:340) and fused with the destination's own estimate by covariance intersection (:362-389).  Nothing is taken from the
     ground truth; the fused position is CHECKED against the pose the frame was rendered from.
     (--scene synthetic keeps round 1's stand-in: the neighbour's error-free relative offset.)
Reports frames/s per camera, per-stage p50 latencies and the position error against ground truth.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rot_y(a):
    import numpy as np
    return np.array([[math.cos(a), 0, -math.sin(a)], [0, 1, 0], [math.sin(a), 0, math.cos(a)]])


def motion_from_essential(E):
    """The four (R, t) candidates of an essential matrix, x2 ~ R x1 + t, |t| = 1 (MotionFromEssential)."""
    import numpy as np
    U, _, Vt = np.linalg.svd(E)
    if np.linalg.det(U) < 0:
        U = -U
    if np.linalg.det(Vt) < 0:
        Vt = -Vt
    Wm = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    R1, R2, t = U @ Wm @ Vt, U @ Wm.T @ Vt, U[:, 2]
    return [(R1, t), (R1, -t), (R2, t), (R2, -t)]


def triangulate_two_views(R, t, n1, n2):
    """Points (camera 1's frame) of normalised image points n1, n2 (n x 2) seen by cameras [I|0] and [R|t]: the depths along the two
    rays that bring them closest (2 x 2 normal equations per point, closed form; the reference's TriangulateDLT differs from it by
    less than the measurement noise and costs a 4 x 4 SVD per point -- 12 ms per pair in numpy against 0.3).  Returns (X1, depth1, depth2)."""
    import numpy as np
    a = np.c_[n1, np.ones(len(n1))] @ R.T            # R [n1; 1]
    b = np.c_[n2, np.ones(len(n2))]
    # min | l1 a - l2 b + t |^2
    aa, bb, ab = (a * a).sum(1), (b * b).sum(1), (a * b).sum(1)
    at, bt = a @ t, b @ t
    det = aa * bb - ab * ab
    det = np.where(np.abs(det) < 1e-18, 1e-18, det)
    l1 = (-at * bb + bt * ab) / det
    l2 = (-at * ab + bt * aa) / det
    X1 = np.c_[n1, np.ones(len(n1))] * l1[:, None]
    return X1, l1, l2


def relative_pose_from_essential(E, K, x1, x2):
    """RelativePoseFromEssential (RobustMatcher.hpp:176-183): the candidate with most points in front of both cameras.
    Returns (R, t, X1 (n x 3, camera-1 frame, unit baseline), in_front mask) or None."""
    import numpy as np
    Ki = np.linalg.inv(K)
    n1 = (np.c_[x1, np.ones(len(x1))] @ Ki.T)[:, :2]
    n2 = (np.c_[x2, np.ones(len(x2))] @ Ki.T)[:, :2]
    best = None
    for R, t in motion_from_essential(E):
        X1, l1, l2 = triangulate_two_views(R, t, n1, n2)
        ok = (l1 > 0) & (l2 > 0)
        if best is None or ok.sum() > best[3].sum():
            best = (R, t, X1, ok)
    return best if best is not None and best[3].sum() >= 8 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--cams", type=int, default=8)
    ap.add_argument("--frames", type=int, default=90)      # 3 s at 30 fps
    ap.add_argument("--map-points", type=int, default=4000)
    ap.add_argument("--scene", choices=["plane", "synthetic"], default="plane")
    ap.add_argument("--pipelined", action="store_true", help="plane scene, one rank: the two-lane loop also below 4 cameras")
    ap.add_argument("--sequential", action="store_true", help="plane scene: the camera-by-camera loop only")
    ap.add_argument("--no-inter-shortcut-run", action="store_true",
                    help="plane scene: skip the second run of the loop that uses the map_index shortcut in the inter-camera step (inter_forms)")
    ap.add_argument("--no-overlap", action="store_true", help="plane scene, frame-batched loop: no pipelining -- a frame's front end is enqueued and waited for inside its own timed region")
    ap.add_argument("--unique-frames", type=int, default=6, help="plane scene: rendered frames per camera (the trajectory loops over them)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed without a launcher: start the ranks as a child torch.distributed.run (before anything here touches the GPU), relay its
        # output and exit code -- the same helper as bench.py
        import bench
        sys.exit(bench.self_launch(args.gpus, script=os.path.abspath(__file__)))

    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from coloc_amd import Context, cov_intersection

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > torch.cuda.device_count():
        raise SystemExit("--gpus %d needs %d GPUs (one camera per GPU, RCCL), this host shows %d; one GPU runs --cams cameras with --gpus 1"
                         % (world, world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    my_cams = list(range(args.cams)) if world == 1 else [rank]
    n_cams = args.cams if world == 1 else world
    if args.scene == "plane":
        plane_scene(args, world, rank, local_rank, dev, my_cams, n_cams)
        if world > 1:
            dist.destroy_process_group()
        return

    W, H, M = 1280, 720, args.map_points
    ctx = Context(device=local_rank, width=W, height=H, maxkp=20000, match_thresh=60)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream

    rng = np.random.default_rng(7)
    Xw = np.stack([rng.uniform(-6, 6, M), rng.uniform(-4, 4, M), rng.uniform(6, 22, M)], 1)     # map landmarks
    map_desc = rng.integers(0, 256, size=(M, 64), dtype=np.uint8)
    ctx.set_map(map_desc)                                                                        # GPUMatcher::setMapData
    K = np.array([[1000.0, 0, W / 2], [0, 1000.0, H / 2], [0, 0, 1]])
    frames = [torch.from_numpy(synth.rect_image(W, H, seed=1000 + c, noise_sigma=2.0)).to(dev) for c in my_cams]

    def pose_of(cam, f):
        """smooth known trajectory: cameras on an arc, slowly orbiting"""
        a = (cam - (n_cams - 1) / 2) * 0.12 + 0.2 * math.sin(2 * math.pi * f / 300.0)
        C = np.array([2.5 * math.sin(a), 0.15 * cam, 2.5 - 2.5 * math.cos(a)])
        R = rot_y(-0.5 * a)
        return R, -R @ C, C

    lat = {"front_end": [], "match": [], "pose": [], "fuse": [], "frame": []}
    pos_err, pos_err_fused, n_inl = [], [], []
    est = {}
    t_start = time.perf_counter()
    for f in range(args.frames):
        for k, cam in enumerate(my_cams):
            t0 = time.perf_counter()
            # 1. front end (device-resident)
            ctx.pyramid_build_dev(frames[k].data_ptr(), W, H, W, sptr)
            ctx.detect_dev(sptr)
            ctx.describe_detected_dev(None, sptr)
            ctx.sync(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            # 2. observed descriptors of this frame: visible landmarks with bit noise + 30 % distractors
            R, t, C = pose_of(cam, f)
            Xc = Xw @ R.T + t
            uv = (Xc @ K.T); uv = uv[:, :2] / uv[:, 2:3]
            vis = np.nonzero((Xc[:, 2] > 0.5) & (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H))[0]
            frng = np.random.default_rng(100000 * cam + f)
            # bit noise: each of the 512 bits flips with probability p_i, p_i ~ U(0, 30/512) per keypoint
            p_flip = frng.uniform(0, 30.0 / 512.0, (len(vis), 1))
            noise = np.packbits(frng.random((len(vis), 512)) < p_flip, axis=1)
            obs = map_desc[vis] ^ noise
            n_dis = int(0.3 * len(vis))
            q = np.concatenate([obs, frng.integers(0, 256, size=(n_dis, 64), dtype=np.uint8)])
            xy = np.concatenate([uv[vis] + frng.normal(0, 0.5, (len(vis), 2)),
                                 np.stack([frng.uniform(0, W, n_dis), frng.uniform(0, H, n_dis)], 1)])
            # in the real loop these descriptors are CLATCH output and already on the device (stage 1): upload the synthetic
            # stand-ins outside the timed stage, then match on the device and bring back only the match indices
            d_q = torch.from_numpy(q).to(dev)
            d_m = torch.empty(q.shape[0], dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            t1b = time.perf_counter()
            ctx.match_map_dev(d_q.data_ptr(), q.shape[0], 60, d_m.data_ptr(), sptr)   # train = map: m[i] = map index or -1  (IndMatch(map, query))
            m = d_m.cpu().numpy()
            t2 = time.perf_counter()
            sel = np.nonzero(m >= 0)[0]
            # 3. robust pose + refinement + covariance
            Rt, cov, mask, rmse = ctx.pnp_localize(Xw[m[sel]], xy[sel], K, n_samples=256, seed=f + 1, thr2=16.0)
            t3 = time.perf_counter()
            if Rt is None:
                continue
            Ce = -Rt[:, :3].T @ Rt[:, 3]
            # covariance of the centre: J = -R^T on the translation block (rotation part neglected, like the
            # reference, which fuses the translation block only: colocUtils.hpp:38-43)
            Cc = Rt[:, :3].T @ cov[3:, 3:] @ Rt[:, :3]
            est[cam] = (Ce, Cc, C)
            lat["front_end"].append(t1 - t0); lat["match"].append(t2 - t1b); lat["pose"].append(t3 - t2)
            lat["frame"].append((t1 - t0) + (t3 - t1b))
            pos_err.append(np.linalg.norm(Ce - C)); n_inl.append(int(mask.sum()))
        # 4. fusion across cameras (positions + 3x3 covariances are tiny: gathered on the host side here)
        if world > 1:
            mine = est.get(rank)
            buf = torch.zeros(15, dtype=torch.float64, device=dev)
            if mine is not None:
                buf[:3] = torch.from_numpy(mine[0]); buf[3:12] = torch.from_numpy(mine[1].reshape(9)); buf[12:15] = torch.from_numpy(mine[2])
            allb = torch.zeros((world, 15), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(allb.view(-1), buf)
            allb = allb.cpu().numpy()
            est = {c: (allb[c, :3], allb[c, 3:12].reshape(3, 3), allb[c, 12:15]) for c in range(world) if allb[c, 3:12].any()}
        for cam in my_cams:
            nb = (cam + 1) % n_cams
            if cam not in est or nb not in est or nb == cam:
                continue
            t4 = time.perf_counter()
            Ce, Cc, Cgt = est[cam]
            Cn, Ccn, Cngt = est[nb]
            rel = Cgt - Cngt                                   # the neighbour's (here: exact) relative measurement
            om, Cf, pf = cov_intersection(Cc, Ccn + 1e-6 * np.eye(3), Ce, Cn + rel)
            lat["fuse"].append(time.perf_counter() - t4)
            pos_err_fused.append(np.linalg.norm(pf - Cgt))
    wall = time.perf_counter() - t_start
    if rank == 0:
        p50 = lambda v: float(np.median(v) * 1e3) if len(v) else None
        out = {"scenario": "config[4]-shaped streaming loop, %d camera(s) on this rank, %d ranks" % (len(my_cams), world),
               "frames_per_camera": args.frames, "map_points": M,
               "camera_frames_per_s_per_gpu_timed_stages": (1000.0 / p50(lat["frame"])) if lat["frame"] else None,
               "cameras_at_30fps_per_gpu": (1000.0 / p50(lat["frame"]) / 30.0) if lat["frame"] else None,
               "wall_s_including_host_side_synthesis": wall,
               "required": "30 fps per camera (config[4]: 8 cameras on 8 GPUs)",
               "p50_ms": {k: p50(v) for k, v in lat.items()},
               "inliers_p50": float(np.median(n_inl)) if n_inl else None,
               "position_error_p50": float(np.median(pos_err)) if pos_err else None,
               "position_error_fused_p50": float(np.median(pos_err_fused)) if pos_err_fused else None,
               "note": "host-side synthesis of the observed descriptors is outside the timed stages"}
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def plane_scene(args, world, rank, local_rank, dev, my_cams, n_cams):
    """config[4] on rendered frames: describe -> match the descriptors just computed against the map -> a-contrario pose +
    refinement -> covariance-intersection fusion.

    One rank with several cameras (world == 1) runs the loop twice over the same frames:
      sequential  round 3's call pattern, camera by camera: pyramid, detect, CLATCH, count to the host, map match, results to the
                  host, pose -- a synchronisation between the stages (the reference's own pattern, coloc.hpp:201-272);
      pipelined   (the headline) two lanes (host thread, context, stream, buffers), lane p owns the cameras p (mod 2): the front end
                  (clc_detect_batch_dev) and the map match with the keypoint count read on the device (clc_match_jobs_counted_dev)
                  of the lane's next camera frame are enqueued -- no host synchronisation in between -- BEFORE the lane solves its
                  current camera frame's pose, and the two lanes' a-contrario chains (short launches with the host in the loop, the
                  GPU mostly idle between them) interleave.  (Batching the front end of all cameras of a
                  frame into one call was measured too: the device time per camera is lower, but eight cameras' kernels beside the
                  latency-bound pose rounds stretched every round: 0.29 -> 0.45 ms per pose.)
    Both produce the same keypoints, matches and poses (checked)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from coloc_amd import Context, cov_intersection
    from coloc_amd.abi import KP_DTYPE

    W, H, PPU, HEIGHT = 1280, 720, 200.0, 5.0
    K = np.array([[1000.0, 0, W / 2], [0, 1000.0, H / 2], [0, 0, 1]])
    CAP = 20000
    ctx = Context(device=local_rank, width=W, height=H, maxkp=CAP, match_thresh=60)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream
    tex = synth.plane_texture(size=2000, seed=77, n_rect=2600)
    centre = np.array([5.0, 5.0])
    # a gentle relief under the cameras: a flat scene leaves the essential matrix with its planar two-fold ambiguity
    relief = synth.smooth_relief(centres=((4.2, 4.4, 0.55, 0.9), (6.0, 5.6, 0.45, 0.7), (5.1, 6.3, 0.40, 0.6), (5.9, 4.0, 0.50, 0.8), (3.6, 6.0, 0.45, 0.7)))

    POW12 = np.power(np.float32(1.2), np.arange(256, dtype=np.float32))        # pow(1.2f, scale), GPUDetector.hpp:173-179

    def feature_xy(kps):
        s = POW12[kps["scale"]]
        out = np.empty((len(kps), 2), dtype=np.float64)
        out[:, 0] = s * kps["x"]
        out[:, 1] = s * kps["y"]
        return out

    # the map: one reference view a little higher up, its CLATCH descriptors + the 3-D points under its keypoints
    Rm, tm = synth.look_at_plane_pose(centre, HEIGHT * 1.06)
    kps_m, desc_m, _ = ctx.detect_and_describe(synth.render_plane(tex, PPU, K, Rm, tm, W, H, relief=relief), capacity=CAP)
    Xmap = synth.backproject_to_plane(feature_xy(kps_m), K, Rm, tm, relief=relief)
    ctx.set_map(desc_m)
    M = len(desc_m)

    pose_cache = {}

    def pose_of(cam, f):
        key = (cam, f % args.unique_frames)
        if key not in pose_cache:
            a = 2 * math.pi * key[1] / args.unique_frames
            off = np.array([0.35 * math.cos(a) + 0.12 * (cam - (n_cams - 1) / 2), 0.25 * math.sin(a)])
            R, t = synth.look_at_plane_pose(centre + off, HEIGHT, yaw=0.08 * (cam - (n_cams - 1) / 2) + 0.05 * math.sin(a),
                                            tilt=(0.03 * math.cos(a), 0.03 * math.sin(a) - 0.01 * cam))
            pose_cache[key] = (R, t, -R.T @ t)
        return pose_cache[key]

    frames = {(cam, u): torch.from_numpy(synth.render_plane(tex, PPU, K, *pose_of(cam, u)[:2], W, H, relief=relief)).to(dev)
              for cam in my_cams for u in range(args.unique_frames)}
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")

    def d2d(dst, src, nbytes, st):
        hip.hipMemcpyAsync(ctypes.c_void_p(dst), ctypes.c_void_p(src), ctypes.c_size_t(nbytes), 3, ctypes.c_void_p(st))

    def new_stats():
        return dict(lat={"front_end": [], "match": [], "pose": [], "inter": [], "fuse": [], "frame": []}, pos_err=[], pos_err_fused=[],
                    pos_err_inter=[], n_inl=[], n_match=[], n_kp=[], n_pair=[], n_pair_inl=[], n_common=[], inter_fail=0, poses={})

    def solve_pose(c, cam, f, n, m, kps, st):
        """stage 3 for one camera: a-contrario P3P + refinement + covariance (Localizer::localizeImage); -> est entry or None"""
        sel = np.nonzero(m >= 0)[0]
        if len(sel) < 8:
            return None
        r = c.pnp_acransac(Xmap[m[sel]], feature_xy(kps[sel]), K, seed=f + 1, refine=True)
        if r["Rt"] is None:
            return None
        Rt, cov = r["Rt"], r["cov"]
        Ce = -Rt[:, :3].T @ Rt[:, 3]
        Cc = Rt[:, :3].T @ cov[3:, 3:] @ Rt[:, :3]
        gt = pose_of(cam, f)[2]
        st["pos_err"].append(np.linalg.norm(Ce - gt)); st["n_inl"].append(len(r["inliers"])); st["n_match"].append(len(sel)); st["n_kp"].append(n)
        st["poses"][(cam, f)] = Rt
        return dict(C=Ce, cov=Cc, gt=gt, Rt=Rt, n=n, kps=kps, m=m)   # feature coordinates are derived for the rows a later stage needs

    inter_pool = {"ctxs": [], "d_pairs": None}
    # how the inter-camera step finds the features the pair's temporary map shares with the global map (clc_inter_pose_batch):
    # "reference" = the reference's own chain (coloc.hpp:317-323: the temporary map's descriptors -- the lower camera's rows of the
    # correspondences -- matched against the global map's on the device, K2NN threshold 60); "map_index" = the shortcut of rounds 3-5
    # (the source frame's own map matches say which of its features the global map holds).  The headline runs the reference's chain,
    # the same loop runs again with the shortcut and both are reported (inter_forms).
    inter_form = {"v": "reference"}
    d_map_desc_t = torch.from_numpy(np.ascontiguousarray(desc_m)).to(dev)

    def inter_and_fuse(c, cptr, f, est, desc_ptr, d_pair, st):
        """inter-camera step + fusion for every camera of the frame: destination = cam, source = its neighbour (coloc.hpp:274-392).
        Round 5: the pairs' match sweeps are enqueued together and fetched behind ONE synchronisation, and everything between the putative
        matches and the covariance intersection -- a-contrario five-point filter, relative pose from E, temporary map, scale, refinement
        -- is ONE clc_inter_pose_batch call for all pairs (rounds 3-4 did that geometry in numpy, pair after pair)."""
        from coloc_amd.abi import inter_pose_batch
        lat = st["lat"]
        todo = []
        for cam in my_cams:
            nb = (cam + 1) % n_cams
            if cam in est and nb in est and nb != cam:
                todo.append((cam, nb))
        if not todo:
            return
        t4 = time.perf_counter()
        if inter_pool["d_pairs"] is None or inter_pool["d_pairs"].shape[0] < len(todo):
            inter_pool["d_pairs"] = torch.empty((len(my_cams), CAP), dtype=torch.int32, device=dev)
        while len(inter_pool["ctxs"]) < len(todo):
            inter_pool["ctxs"].append(Context(device=local_rank, detector=False, matcher=False))
        dp = inter_pool["d_pairs"]
        # computeMatchesPair(source, dest): Q = the source frame's descriptors, T = the destination's (GPUMatcher.hpp:165-172)
        for k, (cam, nb) in enumerate(todo):
            c.match_2nn_dev(desc_ptr[nb], est[nb]["n"], desc_ptr[cam], est[cam]["n"], 60, dp[k].data_ptr(), cptr)
        with torch.cuda.stream(torch.cuda.ExternalStream(cptr)):
            nmax = max(est[nb]["n"] for _, nb in todo)
            block = dp[:len(todo), :nmax].cpu().numpy()              # ONE copy + synchronisation for all pairs of the frame
            pms = [block[k, :est[nb]["n"]] for k, (cam, nb) in enumerate(todo)]
        probs, meta = [], []
        for (cam, nb), pm in zip(todo, pms):
            S, D = est[nb], est[cam]
            q = np.nonzero(pm >= 0)[0]
            if len(q) < 16:
                st["inter_fail"] += 1
                continue
            x1 = feature_xy(S["kps"][q]) if "kps" in S else S["xy"][q]
            x2 = feature_xy(D["kps"][pm[q]]) if "kps" in D else D["xy"][pm[q]]
            # filterMatchesPair + RelativePoseFromEssential + interReconstruct + scale + refinePose (coloc.hpp:296-340); the source frame's
            # map matches (threshold 60) say which of its features the global map holds
            if inter_form["v"] == "reference":
                # setupMapDatabase(inter): a temporary map point keeps the descriptor of its first observation = the camera with the lower id
                lower, rows = (nb, q) if nb < cam else (cam, pm[q])
                probs.append(dict(x1=x1, x2=x2, K=K, wh=(W, H), seed=f + 1, Rt_source=S["Rt"], d_first_desc=desc_ptr[lower],
                                  first_feature=rows.astype(np.int32), d_map_desc=d_map_desc_t.data_ptr(), match_threshold=60))
            else:
                probs.append(dict(x1=x1, x2=x2, K=K, wh=(W, H), seed=f + 1, map_index=S["m"][q], Rt_source=S["Rt"]))
            meta.append((cam, nb, len(q)))
        res = inter_pose_batch(inter_pool["ctxs"][:len(probs)], probs, Xmap) if probs else []
        t5 = time.perf_counter()
        n_ok = 0
        for (cam, nb, nq), r in zip(meta, res):
            S, D = est[nb], est[cam]
            if r["stage"] != 0 or r["status"] != 0:
                st["inter_fail"] += 1
                continue
            Rt_i, cov_i = r["Rt"], r["cov"]
            Ci = -Rt_i[:, :3].T @ Rt_i[:, 3]
            Cci = Rt_i[:, :3].T @ cov_i[3:, 3:] @ Rt_i[:, :3] + S["cov"]     # covInter = currentCov[source] + cov (:366)
            om, Cf, pf = cov_intersection(D["cov"], Cci + 1e-12 * np.eye(3), D["C"], Ci)
            st["pos_err_inter"].append(np.linalg.norm(Ci - D["gt"])); st["pos_err_fused"].append(np.linalg.norm(pf - D["gt"]))
            st["n_pair"].append(nq); st["n_pair_inl"].append(len(r["inliers"])); st["n_common"].append(r["n_common"])
            st.setdefault("inter_centres", {})[(f, cam)] = Ci
            n_ok += 1
        t6 = time.perf_counter()
        for _ in range(n_ok):
            lat["inter"].append((t5 - t4) / len(todo)); lat["fuse"].append((t6 - t5) / max(n_ok, 1))

    # ---------------------------------------------------------------------------------------------------------------------
    # sequential: camera by camera, a synchronisation between the stages (every rank count; the only mode at world > 1)
    # ---------------------------------------------------------------------------------------------------------------------
    def run_sequential(with_inter):
        st = new_stats()
        lat = st["lat"]
        # what a camera's frame leaves behind for the inter-camera step: its descriptors (device), keypoints, map matches, pose + covariance
        cam_desc = {cam: torch.zeros((CAP, 64), dtype=torch.uint8, device=dev) for cam in (my_cams if world == 1 else [rank, (rank + 1) % world])}
        d_pair = torch.empty(CAP, dtype=torch.int32, device=dev)
        d_kps, d_cnt, d_desc = ctx.detect_buffers()
        cnt_view = torch.empty(1, dtype=torch.int32, device=dev)
        d_m = torch.empty(CAP, dtype=torch.int32, device=dev)
        kp_bytes = torch.empty(CAP * KP_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        t_start = time.perf_counter()
        for f in range(args.frames):
            est = {}
            for cam in my_cams:
                img = frames[(cam, f % args.unique_frames)]
                t0 = time.perf_counter()
                # 1. front end, device-resident
                ctx.pyramid_build_dev(img.data_ptr(), W, H, W, sptr)
                ctx.detect_dev(sptr)
                ctx.describe_detected_dev(None, sptr)
                d2d(cnt_view.data_ptr(), d_cnt, 4, sptr)
                n = int(cnt_view.cpu()[0])                       # the keypoint count is the only thing the host needs here
                n = min(n, CAP)
                t1 = time.perf_counter()
                # 2. map tracking on the descriptors stage 1 left on the device
                ctx.match_map_dev(d_desc, n, 60, d_m.data_ptr(), sptr)
                d2d(kp_bytes.data_ptr(), d_kps, n * KP_DTYPE.itemsize, sptr)
                m = d_m[:n].cpu().numpy()
                kps = kp_bytes[:n * KP_DTYPE.itemsize].cpu().numpy().view(KP_DTYPE)
                t2 = time.perf_counter()
                e = solve_pose(ctx, cam, f, n, m, kps, st)
                t3 = time.perf_counter()
                if e is None:
                    continue
                d2d(cam_desc[cam].data_ptr(), d_desc, n * 64, sptr)       # the frame's descriptors stay on the device for the pair match
                est[cam] = e
                lat["front_end"].append(t1 - t0); lat["match"].append(t2 - t1); lat["pose"].append(t3 - t2); lat["frame"].append(t3 - t0)
            if world > 1:
                # one camera per rank: the inter-camera step of camera `rank` (destination) needs the frame of its source, camera
                # rank + 1: descriptors, keypoint coordinates, map matches, pose and covariance travel in fixed-capacity blocks
                mine = est.get(rank)
                blk = torch.zeros((CAP, 7), dtype=torch.float64, device=dev)          # per keypoint: x, y, map match
                hdr = torch.zeros(32, dtype=torch.float64, device=dev)
                if mine is not None:
                    blk[:mine["n"], 0:2] = torch.from_numpy(feature_xy(mine["kps"])).to(dev)
                    blk[:mine["n"], 2] = torch.from_numpy(mine["m"].astype(np.float64)).to(dev)
                    hdr[0] = mine["n"]; hdr[1:13] = torch.from_numpy(mine["Rt"].reshape(12)).to(dev); hdr[13:22] = torch.from_numpy(mine["cov"].reshape(9)).to(dev)
                    hdr[22:25] = torch.from_numpy(mine["gt"]).to(dev); hdr[25] = 1.0
                all_desc = torch.empty((world, CAP, 64), dtype=torch.uint8, device=dev)
                all_blk = torch.empty((world, CAP, 7), dtype=torch.float64, device=dev)
                all_hdr = torch.empty((world, 32), dtype=torch.float64, device=dev)
                dist.all_gather_into_tensor(all_desc.view(-1), cam_desc[rank].view(-1))
                dist.all_gather_into_tensor(all_blk.view(-1), blk.view(-1))
                dist.all_gather_into_tensor(all_hdr.view(-1), hdr)
                src = (rank + 1) % world
                hs = all_hdr[src].cpu().numpy()
                if hs[25] > 0 and src != rank:
                    ns = int(hs[0])
                    bs = all_blk[src, :ns].cpu().numpy()
                    cam_desc[src].copy_(all_desc[src])
                    Rts = hs[1:13].reshape(3, 4)
                    est[src] = dict(C=-Rts[:, :3].T @ Rts[:, 3], cov=hs[13:22].reshape(3, 3), gt=hs[22:25], Rt=Rts, n=ns, xy=bs[:, 0:2].copy(),
                                    m=bs[:, 2].astype(np.int64))
            if with_inter:
                inter_and_fuse(ctx, sptr, f, est, {c: t.data_ptr() for c, t in cam_desc.items()}, d_pair, st)
        st["wall"] = time.perf_counter() - t_start
        return st

    # ---------------------------------------------------------------------------------------------------------------------
    # pipelined (world == 1): two lanes (host thread + context + stream + buffers), lane p owns the cameras b = p (mod 2).  The device
    # work of a lane's next camera frame -- front end and map match, enqueued without any host synchronisation: the keypoint count
    # stays on the device -- is enqueued before the lane's thread starts the pose solve of its current camera frame, and the two lanes'
    # pose solves (host-driven chains of short launches) run side by side
    # ---------------------------------------------------------------------------------------------------------------------
    def run_pipelined():
        st = new_stats()
        lat = st["lat"]
        nc = len(my_cams)
        EAGER = 12288                                        # rows of keypoints / matches copied to the host without knowing the count
        # one arena for the counted map match: block 0 = the map, then the cameras' descriptor blocks of even and of odd frames (the
        # inter-camera step of frame f still reads them while frame f + 1 is being described)
        arena = torch.zeros((1 + 2 * nc, CAP, 64), dtype=torch.uint8, device=dev)
        arena[0, :M] = torch.from_numpy(desc_m).to(dev)
        map_n = torch.tensor([M], dtype=torch.int32, device=dev)
        lanes = []
        for p in range(2):
            c = ctx if p == 0 else Context(device=local_rank, width=W, height=H, maxkp=CAP, match_thresh=60)
            s_ = stream if p == 0 else torch.cuda.Stream(device=dev)
            lanes.append(dict(ctx=c, stream=s_, sptr=s_.cuda_stream,
                              kps=torch.zeros((CAP, 20), dtype=torch.uint8, device=dev), cnt=torch.zeros((2,), dtype=torch.int32, device=dev),
                              match=torch.empty((CAP,), dtype=torch.int32, device=dev),
                              h_cnt=torch.zeros((2,), dtype=torch.int32).pin_memory(), h_kps=torch.zeros((EAGER, 20), dtype=torch.uint8).pin_memory(),
                              h_match=torch.zeros((EAGER,), dtype=torch.int32).pin_memory(), done=torch.cuda.Event(),
                              d_pair=torch.empty(CAP, dtype=torch.int32, device=dev)))
        torch.cuda.synchronize()
        n_items = args.frames * nc

        def block_of(f, b):
            return 1 + (f & 1) * nc + b

        def enqueue(i, p):
            f, b = divmod(i, nc)
            L = lanes[p]
            c, sp = L["ctx"], L["sptr"]
            blk = block_of(f, b)
            c.detect_batch_dev([frames[(my_cams[b], f % args.unique_frames)].data_ptr()], W, H, W, [L["kps"].data_ptr()], [L["cnt"].data_ptr()],
                               [arena[blk].data_ptr()], sp)
            # map tracking (GPUMatcher.hpp:174-178, thr 60) with the count read on the device
            c.match_jobs_counted_dev(arena.data_ptr(), [(blk * CAP, CAP, 0, M, 0, 60)], [L["cnt"].data_ptr()], [map_n.data_ptr()], [0],
                                     L["match"].data_ptr(), sp)
            with torch.cuda.stream(L["stream"]):
                L["h_cnt"].copy_(L["cnt"], non_blocking=True)
                L["h_kps"].copy_(L["kps"][:EAGER], non_blocking=True)
                L["h_match"].copy_(L["match"][:EAGER], non_blocking=True)
                L["done"].record(L["stream"])

        # lane p owns the cameras b = p (mod 2): per camera frame wait -> copy out -> enqueue the lane's NEXT camera frame -> solve
        def items_of(p, f):
            return [f * nc + b for b in range(p, nc, 2)]

        def next_item(p, i):
            f, b = divmod(i, nc)
            if b + 2 < nc:
                return i + 2
            return (f + 1) * nc + p if (f + 1 < args.frames and p < nc) else None

        def lane_frame(p, f):
            L = lanes[p]
            res = []
            for i in items_of(p, f):
                b = i % nc
                t0 = time.perf_counter()
                L["done"].synchronize()
                n = min(int(L["h_cnt"][0]), CAP)
                if n > EAGER:                                # rare: fetch the rows beyond the eager copy
                    with torch.cuda.stream(L["stream"]):
                        m = L["match"][:n].cpu().numpy()
                        kps = L["kps"][:n].cpu().numpy().reshape(-1).view(KP_DTYPE)
                else:
                    m = L["h_match"][:n].numpy().copy()
                    kps = L["h_kps"][:n].numpy().reshape(-1).view(KP_DTYPE).copy()
                nx = next_item(p, i)
                if nx is not None:
                    enqueue(nx, p)                           # the GPU works on the lane's next camera frame while the host solves this pose
                t1 = time.perf_counter()
                e = solve_pose(L["ctx"], my_cams[b], f, n, m, kps, L["stats"])
                t2 = time.perf_counter()
                res.append((my_cams[b], e, t1 - t0, t2 - t1))
            return res

        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=2)
        # (measured and dropped: lane 0 on the main thread with a 50 us interpreter switch interval -- 0.27 -> 0.40 ms per camera frame)
        for p in range(2):
            lanes[p]["stats"] = new_stats()
        # untimed: every lane runs its first camera frame twice (the second context's first launches allocate its workspaces)
        for _ in range(2):
            for p in range(min(2, nc)):
                enqueue(p, p)
            for fu in [pool.submit(lambda q: (lanes[q]["done"].synchronize(), solve_pose(lanes[q]["ctx"], my_cams[q], 0, min(int(lanes[q]["h_cnt"][0]), EAGER),
                                                                                       lanes[q]["h_match"][:min(int(lanes[q]["h_cnt"][0]), EAGER)].numpy().copy(),
                                                                                       lanes[q]["h_kps"][:min(int(lanes[q]["h_cnt"][0]), EAGER)].numpy().reshape(-1).view(KP_DTYPE).copy(),
                                                                                       new_stats())), p) for p in range(min(2, nc))]:
                fu.result()
        t_start = time.perf_counter()
        for p in range(min(2, nc)):
            enqueue(p, p)
        for f in range(args.frames):
            t0 = time.perf_counter()
            futs = [pool.submit(lane_frame, p, f) for p in range(min(2, nc))]
            res = [r for fu in futs for r in fu.result()]
            t1 = time.perf_counter()
            est = {}
            for cam, e, tw, tp in res:
                if e is not None:
                    est[cam] = e
                    lat["front_end"].append(tw); lat["pose"].append(tp); lat["frame"].append((t1 - t0) / nc)
            L = lanes[0]
            inter_and_fuse(L["ctx"], L["sptr"], f, est, {cm: arena[block_of(f, bb)].data_ptr() for bb, cm in enumerate(my_cams)}, L["d_pair"], st)
        pool.shutdown()
        for p in range(2):
            for k in ("pos_err", "n_inl", "n_match", "n_kp"):
                st[k] += lanes[p]["stats"][k]
            st["poses"].update(lanes[p]["stats"]["poses"])
        st["wall"] = time.perf_counter() - t_start
        lanes[1]["ctx"].close()
        return st

    # ---------------------------------------------------------------------------------------------------------------------
    # frame-batched (world == 1, 2 .. 8 cameras): every stage handles ALL cameras of a frame in one call -- one front-end call
    # (clc_detect_batch_dev: one pyramid, two detector and one CLATCH launch for the frame's cameras), one counted map-match launch with
    # a job per camera, one copy-out, ONE batched a-contrario solve (clc_pnp_localize_ac_batch: the cameras' solves, each on a light
    # context of its own, interleave on the device under one host thread) -- and the device work of frame f + 1 is enqueued on a second
    # stream before frame f's poses are solved.  (Round 4's first attempt at frame-granular batching still solved the poses one after the
    # other beside the batched front end; the batched solve is what makes it pay.)
    # ---------------------------------------------------------------------------------------------------------------------
    def run_batched(overlap=True):
        from coloc_amd.abi import pnp_localize_batch
        st = new_stats()
        lat = st["lat"]
        nc = len(my_cams)
        EAGER = 12288
        arena = torch.zeros((1 + 2 * nc, CAP, 64), dtype=torch.uint8, device=dev)
        arena[0, :M] = torch.from_numpy(desc_m).to(dev)
        map_n = torch.tensor([M], dtype=torch.int32, device=dev)
        fe_stream = torch.cuda.Stream(device=dev)
        fe_ctx = Context(device=local_rank, width=W, height=H, maxkp=CAP, match_thresh=60)      # the front end's context (its own workspaces)
        pose_ctxs = [Context(device=local_rank, detector=False, matcher=False) for _ in range(nc)]
        sets = [dict(kps=torch.zeros((nc, CAP, 20), dtype=torch.uint8, device=dev), cnt=torch.zeros((nc, 2), dtype=torch.int32, device=dev),
                     match=torch.empty((nc, CAP), dtype=torch.int32, device=dev),
                     h_cnt=torch.zeros((nc, 2), dtype=torch.int32).pin_memory(), h_kps=torch.zeros((nc, EAGER, 20), dtype=torch.uint8).pin_memory(),
                     h_match=torch.zeros((nc, EAGER), dtype=torch.int32).pin_memory(), done=torch.cuda.Event()) for _ in range(2)]
        d_pair = torch.empty(CAP, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

        def block_of(f, b):
            return 1 + (f & 1) * nc + b

        def enqueue(f):
            S = sets[f & 1]
            sp = fe_stream.cuda_stream
            fe_ctx.detect_batch_dev([frames[(cam, f % args.unique_frames)].data_ptr() for cam in my_cams], W, H, W,
                                    [S["kps"][b].data_ptr() for b in range(nc)], [S["cnt"][b].data_ptr() for b in range(nc)],
                                    [arena[block_of(f, b)].data_ptr() for b in range(nc)], sp)
            # map tracking of every camera (GPUMatcher.hpp:174-178, thr 60) in one sweep launch, the counts read on the device
            fe_ctx.match_jobs_counted_dev(arena.data_ptr(), [(block_of(f, b) * CAP, CAP, 0, M, b * CAP, 60) for b in range(nc)],
                                          [S["cnt"][b].data_ptr() for b in range(nc)], [map_n.data_ptr()] * nc, [0] * nc, S["match"].data_ptr(), sp)
            with torch.cuda.stream(fe_stream):
                S["h_cnt"].copy_(S["cnt"], non_blocking=True)
                S["h_kps"].copy_(S["kps"][:, :EAGER], non_blocking=True)
                S["h_match"].copy_(S["match"][:, :EAGER], non_blocking=True)
                S["done"].record(fe_stream)

        def solve_frame(f, stats, enqueue_next):
            S = sets[f & 1]
            t0 = time.perf_counter()
            if not overlap:
                enqueue(f)                                   # (--no-overlap: the frame's own front end inside its timed region, nothing hidden)
            S["done"].synchronize()
            per_cam = []
            for b in range(nc):
                n = min(int(S["h_cnt"][b, 0]), CAP)
                if n > EAGER:                                # rare: fetch the rows beyond the eager copy
                    with torch.cuda.stream(fe_stream):
                        m = S["match"][b, :n].cpu().numpy()
                        kps = S["kps"][b, :n].cpu().numpy().reshape(-1).view(KP_DTYPE)
                else:
                    # views of the pinned rows: this set is next written for frame f + 2, behind frame f's inter-camera step
                    m = S["h_match"][b, :n].numpy()
                    kps = S["h_kps"][b, :n].numpy().reshape(-1).view(KP_DTYPE)
                per_cam.append((n, m, kps))
            if overlap and enqueue_next:
                enqueue(f + 1)                               # the GPU runs the next frame's front end beside this frame's pose rounds
            t1 = time.perf_counter()
            jobs, who = [], []
            for b, (n, m, kps) in enumerate(per_cam):
                sel = np.nonzero(m >= 0)[0]
                if len(sel) >= 8:
                    jobs.append((Xmap[m[sel]], feature_xy(kps[sel]), K)); who.append((b, sel))
            res = pnp_localize_batch(pose_ctxs[:len(jobs)], jobs, max_iteration=256, seeds=[f + 1] * len(jobs), refine=True) if jobs else []
            est = {}
            for (b, sel), r in zip(who, res):
                if r["Rt"] is None:
                    continue
                cam = my_cams[b]
                n, m, kps = per_cam[b]
                Rt, cov = r["Rt"], r["cov"]
                Ce = -Rt[:, :3].T @ Rt[:, 3]
                Cc = Rt[:, :3].T @ cov[3:, 3:] @ Rt[:, :3]
                gt = pose_of(cam, f)[2]
                stats["pos_err"].append(np.linalg.norm(Ce - gt)); stats["n_inl"].append(len(r["inliers"])); stats["n_match"].append(len(sel)); stats["n_kp"].append(n)
                stats["poses"][(cam, f)] = Rt
                est[cam] = dict(C=Ce, cov=Cc, gt=gt, Rt=Rt, n=n, kps=kps, m=m)
            t2 = time.perf_counter()
            return est, t1 - t0, t2 - t1

        # untimed: the first frame twice (the contexts allocate their workspaces on their first calls)
        for _ in range(2):
            if overlap:
                enqueue(0)
            solve_frame(0, new_stats(), False)
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        if overlap:
            enqueue(0)
        for f in range(args.frames):
            t0 = time.perf_counter()
            est, tw, tp = solve_frame(f, st, f + 1 < args.frames)
            t1 = time.perf_counter()
            for cam in est:
                lat["front_end"].append(tw / nc); lat["pose"].append(tp / nc); lat["frame"].append((t1 - t0) / nc)
            inter_and_fuse(ctx, sptr, f, est, {cm: arena[block_of(f, bb)].data_ptr() for bb, cm in enumerate(my_cams)}, d_pair, st)
        st["wall"] = time.perf_counter() - t_start
        torch.cuda.synchronize()
        fe_ctx.close()
        for c in pose_ctxs:
            c.close()
        return st

    # two lanes pay off once each has a chain of camera frames per frame to overlap: measured 0.42 -> 0.27-0.29 ms per camera frame at
    # 8 cameras, no gain at 2 or 3 (the per-frame hand-over between the host threads costs what the overlap buys)
    pipelined = world == 1 and (len(my_cams) >= 4 or args.pipelined) and not args.sequential
    batched = world == 1 and 2 <= len(my_cams) <= 8 and not args.sequential and not args.pipelined
    seq = run_sequential(with_inter=not (pipelined or batched))
    st = seq
    same = True
    lanes_st = None
    if pipelined:
        st = lanes_st = run_pipelined()
        # the two call patterns must have found the same poses
        same = all(np.array_equal(seq["poses"][k], st["poses"].get(k)) for k in seq["poses"]) and len(seq["poses"]) == len(st["poses"])
    if batched:
        st = run_batched(overlap=not args.no_overlap)
        same = same and all(np.array_equal(seq["poses"][k], st["poses"].get(k)) for k in seq["poses"]) and len(seq["poses"]) == len(st["poses"])
    # the same loop once more with the map_index shortcut in the inter-camera step (the headline above ran the reference's chain)
    st_short = None
    if not args.no_inter_shortcut_run:
        inter_form["v"] = "map_index"
        st_short = run_batched(overlap=not args.no_overlap) if batched else (run_pipelined() if pipelined else run_sequential(with_inter=True))
        inter_form["v"] = "reference"
    if rank == 0:
        p50 = lambda v: float(np.median(v) * 1e3) if len(v) else None
        med = lambda v: float(np.median(v)) if len(v) else None
        lat = st["lat"]
        pos_err, pos_err_fused, pos_err_inter = st["pos_err"], st["pos_err_fused"], st["pos_err_inter"]
        out = {"scenario": "config[4]-shaped streaming loop on RENDERED frames (textured plane, %d x %d), %d camera(s) on this rank, %d ranks"
                           % (W, H, len(my_cams), world),
               "mode": ("frame-batched: one front-end call, one counted map-match launch and ONE batched a-contrario solve (clc_pnp_localize_ac_batch) for "
                        "all cameras of a frame, the next frame's device work enqueued on a second stream before the poses are solved; frame = a "
                        "frame's wall time / cameras") if batched else
                       ("pipelined: two lanes (host thread + context + stream), each enqueues the front end + counted map match of its next "
                        "camera frame (no host synchronisation) before it solves its current frame's pose; frame = a frame's wall time over "
                        "both lanes / cameras" if pipelined else "sequential: camera by camera, a synchronisation between the stages"),
               "frames_per_camera": args.frames, "localized_frames": len(pos_err), "map_points": int(len(desc_m)),
               "camera_frames_per_s_per_gpu": (1000.0 / p50(lat["frame"])) if lat["frame"] else None,
               "cameras_at_30fps_per_gpu": (1000.0 / p50(lat["frame"]) / 30.0) if lat["frame"] else None,
               "camera_frames_per_s_incl_inter": (len(pos_err) / st["wall"]) if st.get("wall") else None,
               "cameras_at_30fps_per_gpu_incl_inter": (len(pos_err) / st["wall"] / 30.0) if st.get("wall") else None,
               "throughput_what": "camera_frames_per_s_per_gpu / cameras_at_30fps_per_gpu = 1 / the p50 of a camera frame's front end + map match + "
                                  "pose (the intra-camera stages); *_incl_inter = localized camera frames / the loop's wall time, i.e. WITH the "
                                  "inter-camera step and the fusion of every frame -- the figure for the loop BASELINE config[4] names",
               "wall_s": st["wall"], "required": "30 fps per camera (config[4]: 8 cameras on 8 GPUs)",
               "p50_ms": {k: p50(v) for k, v in lat.items()},
               "keypoints_p50": med(st["n_kp"]), "map_matches_p50": med(st["n_match"]), "inliers_p50": med(st["n_inl"]),
               "position_error_p50": med(pos_err), "position_error_max": float(np.max(pos_err)) if pos_err else None,
               "position_error_fused_p50": med(pos_err_fused), "position_error_fused_max": float(np.max(pos_err_fused)) if pos_err_fused else None,
               "position_error_inter_p50": med(pos_err_inter), "position_error_inter_max": float(np.max(pos_err_inter)) if pos_err_inter else None,
               "inter_steps": len(pos_err_inter), "inter_failures": st["inter_fail"],
               "pair_matches_p50": med(st["n_pair"]), "pair_inliers_p50": med(st["n_pair_inl"]), "common_map_features_p50": med(st["n_common"]),
               "inter_rule": "frame-to-frame K2NN (thr 60) -> a-contrario five-point -> relative pose from E -> scale from the features the "
                             "temporary and the global map share -> LM refinement of the destination pose -> covariance intersection "
                             "(coloc.hpp:274-392); nothing taken from the rendered poses; since round 5 one clc_inter_pose_batch call per "
                             "frame for all pairs (p50_ms.inter = that call + the pairs' match sweeps, per pair)",
               "camera_height": HEIGHT,
               "pose_rule": "a-contrario P3P, 256 iterations, error_max = inf (Localizer.hpp:82-93) + LM refinement",
               "note": "frames are rendered on the host before the loop; everything from the uploaded frame to the fused position is timed"}
        def form_summary(s_):
            return {"camera_frames_per_s_incl_inter": (len(s_["pos_err"]) / s_["wall"]) if s_.get("wall") else None,
                    "cameras_at_30fps_per_gpu_incl_inter": (len(s_["pos_err"]) / s_["wall"] / 30.0) if s_.get("wall") else None,
                    "inter_p50_ms": p50(s_["lat"]["inter"]), "position_error_inter_p50": med(s_["pos_err_inter"]),
                    "position_error_inter_max": float(np.max(s_["pos_err_inter"])) if s_["pos_err_inter"] else None,
                    "position_error_fused_p50": med(s_["pos_err_fused"]), "common_map_features_p50": med(s_["n_common"]),
                    "inter_steps": len(s_["pos_err_inter"]), "inter_failures": s_["inter_fail"]}
        out["inter_forms"] = {"reference_chain": dict(form_summary(st), what="the temporary map's descriptors (the lower camera's rows of the pair's "
                                                      "correspondences, gathered on the device) matched against the global map's: K2NN, Q = map, T = temporary map, "
                                                      "threshold 60 -- setupMapDatabase(inter) + matchMapFeatures, coloc.hpp:317-323; THE HEADLINE")}
        if st_short is not None:
            out["inter_forms"]["map_index_shortcut"] = dict(form_summary(st_short), what="the source frame's own map matches say which of its features the global "
                                                            "map holds (rounds 3-5); no descriptor work in the inter-camera step")
            a, b = st.get("inter_centres", {}), st_short.get("inter_centres", {})
            both = sorted(set(a) & set(b))
            dd = [float(np.linalg.norm(a[k] - b[k])) for k in both]
            out["inter_forms"]["centres_compared"] = len(both)
            out["inter_forms"]["centre_difference_p50"] = med(dd)
            out["inter_forms"]["centre_difference_max"] = float(np.max(dd)) if dd else None
            out["inter_forms"]["tolerance"] = ("the two forms walk (nearly) the same common features in different orders -- by global map index, by "
                                               "correspondence -- through the same scale rule (mean of consecutive distance ratios, colocUtils.hpp:184-211): "
                                               "centres agree to that rule's noise -- checked: the MEDIAN difference < 0.01 x camera height; single frames can differ by more, "
                                               "where one form's consecutive-ratio mean meets a bad pair of neighbours (see each form's position_error_inter_max against ground truth)")
        if pipelined or batched:
            sl = seq["lat"]
            out["sequential"] = {"what": "the same frames camera by camera with a synchronisation between the stages (round 3's call pattern), "
                                         "intra-camera stages only", "p50_ms": {k: p50(v) for k, v in sl.items() if v},
                                 "cameras_at_30fps_per_gpu": (1000.0 / p50(sl["frame"]) / 30.0) if sl["frame"] else None,
                                 "localized_frames": len(seq["pos_err"]), "wall_s": seq["wall"]}
            out["same_poses_both_modes"] = bool(same)
            out["frame_p50_ms"] = {"sequential": p50(sl["frame"])}
            if lanes_st is not None:
                ll = lanes_st["lat"]
                out["frame_p50_ms"]["pipelined"] = p50(ll["frame"])
                out["two_lanes"] = {"what": "two lanes (host thread + context + stream), poses solved one after the other inside a lane",
                                    "p50_ms": {k: p50(v) for k, v in ll.items() if v},
                                    "cameras_at_30fps_per_gpu": (1000.0 / p50(ll["frame"]) / 30.0) if ll["frame"] else None}
            if batched:
                out["frame_p50_ms"]["batched"] = p50(lat["frame"])
        print(json.dumps(out))
        if not same:
            sys.exit(3)
        if out["inter_forms"].get("centre_difference_p50") is not None and out["inter_forms"]["centre_difference_p50"] > 0.01 * HEIGHT:
            sys.exit(4)
    for c_ in inter_pool["ctxs"]:
        c_.close()
    ctx.close()


if __name__ == "__main__":
    main()
