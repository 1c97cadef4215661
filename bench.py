#!/usr/bin/env python3
"""bench.py -- CoLoC hot path on MI355X: describe (pyramid + CLATCH) -> 512-bit Hamming 2-NN match.

Metric (BASELINE.json): Mmatches/s = 512-bit Hamming comparisons per second / 1e6 for the all-pairs
sweep at 10k keypoints per image (value), with Mdesc/s and the pose-solve latency reported
beside it.  One step = one pass of the hot path over one batch of synthetic input that is already
resident in HBM:
  N = 1 : BASELINE config[1] -- 2 images (640x480) x 10k keypoints: 2 x (pyramid + CLATCH) and the
          one pair's 10k x 10k K2NN sweep (Q = image 0, T = image 1, threshold 40).
  N > 1 : BASELINE config[3] -- one camera per GPU: pyramid + CLATCH of the rank's own image, RCCL
          all-gather of the 10k x 64 B descriptor block, then the rank's share of the N(N-1)/2 pair
          sweeps (coloc_amd/multicam.py).  Per-GPU describe work is fixed ("weak").
Launch for N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

The cameras look at the SAME synthetic scene (one rectangle image, independent sensor noise per camera, the same
keypoints in a camera-specific order), so the sweep's accept branch does real work and "accepted matches" means
something.  Besides the contract's K timed steps the line carries a sustained leg (>= 1 s of back-to-back steps) and
the shader clock measured inside the sweep kernel right after it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, NKP, THR = 640, 480, 10000, 40
CLOCK_GHZ = 2.4
# K2NN sweep, matrix formulation: 512 multiply-adds per comparison on v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands;
# dense FP4 peak = 256 CU x 4 SIMD x (32 x 32 x 64 x 2 flop / 32 cycles) x 2.4 GHz (MI355X_MICROARCH.md: ~10 PF dense)
MFMA_FP4_PEAK_TFLOPS = 256 * 4 * (32 * 32 * 64 * 2 / 32) * CLOCK_GHZ * 1e9 / 1e12
# popcount formulation (A/B reference): v_bcnt_u32_b32 / SGPR-operand VALU ops issue one wave64 instruction per 4 cycles
VALU_PEAK_TLANEOPS = 256 * 64 * CLOCK_GHZ * 1e9 / 1e12
HBM_PEAK_GBS = 8000.0


def cpu_baseline(desc_q, desc_t, xy_q, xy_t):
    """The oracle's OpenMP brute-force matcher on the same 10k x 10k pair, all host cores; the CHECKER timed as a
    baseline, never the product path.  `value` is the loop BASELINE.md section 2 specifies (restated OpenMVG
    BRUTE_FORCE_HAMMING: 8 x __builtin_popcountll per pair, running top-2, OpenMP over queries) with the K2NN
    acceptance rule; the same loop with OpenMVG's distance-ratio rule (CPUMatcher.hpp:67-76: DistanceRatioMatch(0.8))
    and the best-effort AVX-512 VPOPCNTDQ inner loop are timed beside it."""
    import oracle_lib
    orc = oracle_lib.Oracle()
    reps = 7

    def best_of(kernel, rule):
        # `reps` sweeps inside ONE parallel region, each timed between two team barriers: the team's wake-up is not in the figure
        _, nthr, best = orc.k2nn_omp_timed(desc_q, desc_t, rule=rule, threshold=THR, ratio=0.8, kernel=kernel, reps=reps)
        return best, nthr

    n_cmp = desc_q.shape[0] * desc_t.shape[0]
    t_scalar, nthr = best_of(0, 0)
    t_ratio, n_ratio = None, 0
    for _ in range(reps):                                   # computeMatchesPair(pair): regions[first] = database, [second] = queries
        t0 = time.perf_counter()
        pairs, _ = orc.cpumatcher_pair(desc_q, xy_q, desc_t, xy_t, ratio=0.8, kernel=0)
        dt = time.perf_counter() - t0
        t_ratio, n_ratio = (dt if t_ratio is None else min(t_ratio, dt)), int(pairs.shape[0])
    out = {"value": n_cmp / t_scalar / 1e6, "unit": "Mmatches/s", "cores": int(nthr), "kind": "port",
           "sample": "full %d x %d pair, K2NN acceptance rule, best of %d sweeps inside one parallel region (%.4f s), 8 x popcount64 per pair (BASELINE.md plan)"
                     % (desc_q.shape[0], desc_t.shape[0], reps, t_scalar),
           "cpu_count": os.cpu_count(),
           "openmvg_ratio_rule": {"value": n_cmp / t_ratio / 1e6, "unit": "Mmatches/s",
                                  "matches": n_ratio,
                                  "what": "the whole CPUMatcher::computeMatchesPair restated (CPUMatcher.hpp:67-76: DistanceRatioMatch(0.8), "
                                          "database = first camera, IndMatch(db, query), both de-duplication passes), same popcount loop, "
                                          "best of %d (%.4f s)" % (reps, t_ratio)}}
    if orc.avx512_available():
        t_simd, _ = best_of(1, 0)
        out["best_effort_simd"] = {"value": n_cmp / t_simd / 1e6, "unit": "Mmatches/s",
                                   "what": "same matcher, AVX-512 VPOPCNTDQ transposing inner loop (%.4f s)" % t_simd}
    return out


def self_launch(n_ranks, script=None):
    """`python bench.py --gpus N` typed as it stands (the form of the driver's N = 1 command): start the N ranks as a CHILD
    `python -m torch.distributed.run` with the same arguments, let it write to this process's stdout / stderr, leave with its exit
    code.  Runs before anything here has imported torch or touched the GPU (a process that has initialised the GPU must never be
    replaced, and need not be: this one only waits)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL / peer mappings across processes)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench: --gpus %d without WORLD_SIZE: starting the ranks myself: %s\n" % (n_ranks, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def run_exchange_legs(args, ctx, dist, torch, dev, world, rank, step, fence, d_match, n_out, mine, arena, counts,
                      img_ptrs, kp_ptrs, sptr, mc_headline, fallback_line):
    """bench.py --gpus N, after the headline measurement: the same step through clc_mc_gather_enqueue_dev + clc_mc_match_enqueue_dev
    with the RCCL all-gather (clc-rccl) and with IPC peer copies (clc-peer); each leg's matches are compared with the headline
    exchange's, each is timed over 20 steps (max over ranks).  With --backend gloo (ranks sharing a GPU: no communicator possible) the
    legs run on rehearsal handles -- the other ranks' blocks are filed from the gathered arena -- so that this control flow is covered
    by the CPU-launchable rehearsal.  Returns {leg: {us_per_step, identical} | {error}}."""
    import threading
    from coloc_amd import MultiCam
    real = args.backend == "nccl"
    flag_dev = dev if real else "cpu"
    out = {}

    def all_ok(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=flag_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def watchdog():
        sys.stderr.write("bench.py: an exchange leg did not return within its limit (rank %d); legs so far: %s\n" % (rank, json.dumps(out)))
        sys.stderr.flush()
        if rank == 0:          # the headline was measured (with the torch exchange) before the legs: it is not lost with them
            print(json.dumps(dict(fallback_line, exchange_legs=dict(out, error="an exchange leg did not return within its limit"),
                                  collective_fallback="an exchange leg of the product's clc_mc_* path did not return within %s s: the torch "
                                                      "exchange's measurement stands" % os.environ.get("BENCH_LEG_LIMIT", "90"),
                                  section_errors=["exchange_legs"])), flush=True)
        # every rank's watchdog fires (each is stuck in, or waiting for, the same leg).  The line above is complete -- the failure is IN it
        # (section_errors, collective_fallback) -- but a leg that hangs in a process that has touched the GPU (an RCCL or peer-copy
        # deadlock, a wedged device) is a FAILURE of the run and leaves as one: exit code 4 (BENCH_LEG_HANG_RC overrides; a launcher that
        # wants the line parses it separately from the code)
        os._exit(int(os.environ.get("BENCH_LEG_HANG_RC", "4")))

    # the reference result: one more step of the headline exchange
    step()
    fence()
    want = d_match[:n_out].clone()
    # clc-rccl-overlap (round 6): the same exchange with step k + 1's describe + exchange on one stream and step k's sweep on another
    # (clc_mc_set_overlap: three arena buffers, two event chains); same matches, timed the same way
    for leg, mode in (("clc-rccl", 0), ("clc-peer", 1), ("clc-rccl-overlap", 0)):
        overlap = leg.endswith("-overlap")
        if mc_headline is not None and args.exchange == leg:
            out[leg] = {"note": "this is the headline exchange of this run"}
            continue
        timer = threading.Timer(float(os.environ.get("BENCH_LEG_LIMIT", "90")), watchdog)
        timer.daemon = True
        timer.start()
        mcx, err = None, None
        box = [None]
        if real:
            if rank == 0:
                try:
                    box[0] = MultiCam.unique_id()
                except Exception as exc:                 # e.g. no librccl to dlopen: every rank must still leave the broadcast
                    box[0] = "unique_id: " + repr(exc)
            dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], str):
            err = box[0]
        else:
            try:
                mcx = MultiCam(ctx, world=world, rank=rank, maxkp=NKP, unique_id=box[0])
            except Exception as exc:
                err = "create: " + repr(exc)
        if err is None and overlap:
            try:
                mcx.set_overlap(True)
            except Exception as exc:
                err = "set_overlap: " + repr(exc)
        if not all_ok(err is None):
            out[leg] = {"error": err or "another rank could not create its handle"}
        else:
            if real and mode == 1:
                try:
                    mcx.open_peers(stream=sptr)
                except Exception as exc:
                    err = "open_peers: " + repr(exc)
            if not all_ok(err is None):
                out[leg] = {"error": err or "another rank could not map the peers' arenas"}
            else:
                got = torch.full_like(d_match, -9)
                torch.cuda.synchronize(dev)      # (torch fills on ITS stream; the library launches on another, non-blocking one: order them)
                sweep_stream = torch.cuda.Stream(device=dev) if overlap else None
                sweep_ptr = sweep_stream.cuda_stream if overlap else sptr
                # what RCCL says about the communicator each rank holds (ncclCommCount, ncclCommUserRank): (0, -1) on rehearsal handles
                try:
                    info = mcx.comm_info()
                except Exception as exc:
                    info = "comm_info: " + repr(exc)
                infos = [None] * world
                dist.all_gather_object(infos, info)

                def leg_step():
                    ctx.describe_batch_dev(img_ptrs, W, H, W, kp_ptrs, [NKP], [mine.data_ptr()], sptr)
                    if not real:
                        for o in range(world):
                            if o != rank:
                                mcx.virtual_put(o, arena[o].data_ptr(), counts[o], stream=sptr)
                    mcx.gather_enqueue_dev(mine.data_ptr(), NKP, mode=mode, stream=sptr)
                    mcx.match_enqueue_dev(THR, got.data_ptr(), got.numel(), stream=sweep_ptr)
                dtk, pf = 0.0, None
                try:
                    for _ in range(3):
                        leg_step()
                    fence()
                    same = bool(torch.equal(want, got[:n_out]))
                    # the contract's measurement for this exchange: W warm-up steps, then EXACTLY K steps between two fences (barrier +
                    # device synchronisation on both sides), the sweep bracketed by HIP events on its stream -- what the headline's
                    # region does with the torch exchange, so that this leg can stand in for it (main(): promotion)
                    for _ in range(args.warmup):
                        leg_step()
                    fence()
                    ctx.profile_reset()
                    ctx.profile_enable(True, only=["k2nn_sweep_kernel"])
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        leg_step()
                    fence()
                    dtk = time.perf_counter() - t0
                    ctx.profile_enable(False)
                    pf = ctx.profile_read()["k2nn_sweep_kernel"]
                    us = dtk / args.steps * 1e6
                    same = same and bool(torch.equal(want, got[:n_out]))
                except Exception as exc:
                    err, same, us, dtk = "step: " + repr(exc), False, 0.0, 0.0
                    try:
                        ctx.profile_enable(False)
                    except Exception:
                        pass
                t = torch.tensor([dtk, 0.0 if same else 1.0, 1.0 if err else 0.0], dtype=torch.float64, device=flag_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                out[leg] = {"us_per_step": float(t[0].item()) / args.steps * 1e6, "identical": bool(t[1].item() == 0.0),
                            "steps": args.steps, "seconds_max_over_ranks": float(t[0].item()), "sweep_prof_rank0": list(pf) if pf else None,
                            "exchange": ("ncclAllGather" if mode == 0 else "IPC peer copies + 4-byte fence all-gather") if real
                                        else "REHEARSAL handle (blocks filed with clc_mc_virtual_put)",
                            "rccl_ranks": infos,           # per rank: [ncclCommCount, ncclCommUserRank] of the handle's communicator
                            "overlapped_steps": overlap}
                if err:
                    out[leg]["error"] = err
                elif t[2].item() != 0.0:
                    out[leg]["error"] = "another rank failed inside its steps"
        if mcx is not None:
            mcx.close()
        timer.cancel()
    return out


def main():
    if os.environ.get("BENCH_FAULT_AFTER"):          # debugging aid: dump every thread's stack and exit if the run takes this long
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_FAULT_AFTER"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle-steps", type=int, default=1000,
                    help="untimed steps BEFORE the W warm-up steps, so that the device clocks have settled when the timed region starts "
                         "(measured: a timed region of 20 steps behind 5 or 50 warm-up steps runs at 114 us/step, behind 500 or 3000 at "
                         "100 us/step -- the clocks of an idle MI355X take 5-50 ms of load to come up).  Reported in the JSON line.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=1.2, help="length of the sustained leg (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real path). gloo = REHEARSAL of the N>1 code path on a box with fewer "
                         "GPUs than ranks: ranks share GPUs and the all-gather is staged through host memory")
    ap.add_argument("--per-camera-launches", action="store_true",
                    help="describe each camera with its own pyramid + CLATCH launch pair (the reference's call pattern) "
                         "instead of the batched entry point")
    ap.add_argument("--exchange", default="auto", choices=["auto", "torch", "clc-rccl", "clc-peer"],
                    help="N > 1 only.  auto (default): the step is measured with torch.distributed's all_gather_into_tensor FIRST (a complete "
                         "line that cannot be lost), then with the PRODUCT's exchange -- clc_mc_gather_enqueue_dev + clc_mc_match_enqueue_dev over "
                         "ncclAllGather -- checked against the torch step's matches; when it runs and agrees on every rank its K timed steps "
                         "become the headline (the torch figure stays in `torch_exchange`), otherwise the torch measurement stands and "
                         "`collective_fallback` says why.  torch: never promote.  clc-rccl / clc-peer: drive the product's exchange from the "
                         "first step on (no fallback).")
    ap.add_argument("--overlap-steps", action="store_true",
                    help="N > 1: promote the OVERLAPPED form of the product's exchange (clc_mc_set_overlap: step k + 1's describe + exchange "
                         "beside step k's sweep, leg clc-rccl-overlap) to the headline instead of the one-stream form; default off")
    ap.add_argument("--no-exchange-legs", action="store_true",
                    help="N > 1 only: skip the informational legs that run the same step through clc-rccl and clc-peer and compare the matches")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph (N=1 only; no in-region events; implies --one-stream)")
    ap.add_argument("--lanes", type=int, default=3, help="lanes of the pipelined N = 1 headline loop (contexts / streams taking consecutive steps in turn; measured 2 / 3 / 4 lanes: 0.0826 / 0.0804-0.0812 / 0.0810 ms per step)")
    ap.add_argument("--one-stream", action="store_true",
                    help="N = 1: every step on ONE stream, a step's sweep waiting for its own describe and the next step's describe for that sweep "
                         "(rounds 1-4's headline).  Default since round 5: consecutive steps are dealt to --lanes contexts / streams / descriptor "
                         "arenas in turn, so that step i's sweep (matrix pipe) runs beside the next steps' pyramid + CLATCH -- the only describe / sweep overlap "
                         "this machine allows (profiles/r05_step_overlap.txt); the one-stream figure is then reported as `one_stream`.")
    ap.add_argument("--sections", default="", help="comma list: run only these side sections (e.g. policy_path,front_end); default all")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the informational side sections (other formulation, shares, config[2], front end, pose, two-view, host path): "
                         "the rocprofv3 passes use it so that a kernel's average in their summaries is the average of the step's own launches")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    import synth
    from coloc_amd import Context, multicam

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or unset WORLD_SIZE and let bench.py start its ranks)"
                         % (args.gpus, world, args.gpus))
    if args.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("--gpus %d with the RCCL backend needs %d GPUs, this host shows %d (--backend gloo rehearses the N > 1 path on fewer)"
                         % (world, world, torch.cuda.device_count()))
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    ctx = Context(device=dev_index, width=W, height=H, maxkp=NKP)
    formulation = os.environ.get("CLC_K2NN_FORMULATION", "matrix")
    formulation = "popcount" if formulation[:1] in ("p", "1") else "matrix"
    # everything (our kernels and torch.distributed's collectives) is ordered on ONE explicit stream
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sptr = stream.cuda_stream

    # ---- synthetic inputs, resident in HBM before the timed region --------------------------
    # one scene; camera c = the scene + its own sensor noise, and the scene's keypoints in its own order
    cams = [0, 1] if world == 1 else [rank]
    scene = synth.rect_image(W, H, seed=1000, noise_sigma=0.0).astype(np.float32)
    base_kps = synth.random_keypoints(NKP, W, H, seed=2000)

    def camera_image(c):
        rng = np.random.default_rng(1100 + c)
        return np.clip(scene + rng.normal(0.0, 2.0, scene.shape) + 0.5, 0, 255).astype(np.uint8)

    def camera_keypoints(c):
        return base_kps[np.random.default_rng(2100 + c).permutation(NKP)]

    imgs = [torch.from_numpy(camera_image(c)).to(dev) for c in cams]
    kps_np = [camera_keypoints(c) for c in cams]
    kps = [torch.from_numpy(k.view(np.uint8).reshape(-1, 20).copy()).to(dev) for k in kps_np]
    n_cams = 2 if world == 1 else world
    counts = [NKP] * n_cams
    arena = torch.zeros((n_cams, NKP, 64), dtype=torch.uint8, device=dev)
    mine = torch.zeros((NKP, 64), dtype=torch.uint8, device=dev)      # this rank's block (all-gather input)
    jobs = multicam.shard_pairs(counts, world, rank, grain=ctx.k2nn_queries_per_block)
    abi_jobs = multicam.jobs_to_abi(jobs, counts, NKP, THR)
    n_out = sum(j.nq for j in jobs)
    d_match = torch.empty((max(n_out, 1),), dtype=torch.int32, device=dev)
    my_cmp = sum(j.nq * counts[j.pair[1]] for j in jobs)
    total_cmp = sum(counts[i] * counts[j] for i, j in multicam.exhaustive_pairs(n_cams))

    # the frames of all cameras this rank owns go through ONE pyramid launch and ONE CLATCH launch
    # (clc_describe_batch_dev; at N > 1 a rank owns one camera)
    img_ptrs = [t.data_ptr() for t in imgs]
    kp_ptrs = [t.data_ptr() for t in kps]
    desc_ptrs = [(arena[c] if world == 1 else mine).data_ptr() for c in cams]

    mc = None
    if world > 1 and args.exchange in ("clc-rccl", "clc-peer") and args.backend == "nccl":
        from coloc_amd import MultiCam
        box = [MultiCam.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        mc = MultiCam(ctx, world=world, rank=rank, maxkp=NKP, unique_id=box[0])
        mc_mode = 0 if args.exchange == "clc-rccl" else 1

    def step():
        if mc is not None:
            ctx.describe_batch_dev(img_ptrs, W, H, W, kp_ptrs, [NKP], [mine.data_ptr()], sptr)
            # enqueue only: no host synchronisation between the exchange and the sweep (capacity-planned shares, the sweep
            # reads the gathered counts on the device; the arena is double-buffered by step parity)
            mc.gather_enqueue_dev(mine.data_ptr(), NKP, mode=mc_mode, stream=sptr)
            mc.match_enqueue_dev(THR, d_match.data_ptr(), d_match.numel(), stream=sptr)
            return
        if args.per_camera_launches:
            for k, c in enumerate(cams):
                ctx.pyramid_build_dev(img_ptrs[k], W, H, W, sptr)
                ctx.describe_dev(kp_ptrs[k], NKP, desc_ptrs[k], sptr)
        else:
            ctx.describe_batch_dev(img_ptrs, W, H, W, kp_ptrs, [NKP] * len(cams), desc_ptrs, sptr)
        if world > 1:
            if args.backend == "nccl":
                dist.all_gather_into_tensor(arena.view(-1), mine.view(-1))   # RCCL over xGMI, 640 KB per rank
            else:                                                            # rehearsal: staged through the host
                host = [torch.empty((NKP, 64), dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(host, mine.cpu())
                arena.copy_(torch.stack(host).to(dev))
        if abi_jobs:
            ctx.match_jobs_dev(arena.data_ptr(), abi_jobs, d_match.data_ptr(), sptr)

    # N = 1 headline loop (round 5): consecutive steps dealt to the lanes in turn; lane = (context, stream, descriptor arena, match buffer).  A lane's
    # step is the whole step -- pyramid + CLATCH of both cameras, then the pair's sweep, in stream order --; what changes against ONE lane is
    # that the device is not drained between step i's sweep and step i + 1's describe: the sweep, dispatched first, holds 58 KB of a CU's
    # LDS and the matrix pipe, the next step's CLATCH waves fill the rest (profiles/r05_step_overlap.txt: inside one step the same overlap
    # is impossible).  One context per concurrently running stream is the library's rule (include/coloc_hip.h).
    pipelined = world == 1 and not args.one_stream and not args.graph and not args.per_camera_launches
    extra_lanes = []                        # (context, torch stream, arena, match buffer) of lanes 1 ..
    if pipelined:
        lanes = [(ctx, sptr, arena, d_match, desc_ptrs)]
        for _ in range(max(2, min(args.lanes, 6)) - 1):
            c_x = Context(device=dev_index, width=W, height=H, maxkp=NKP)
            s_x = torch.cuda.Stream(device=dev)
            a_x = torch.zeros_like(arena)
            m_x = torch.empty_like(d_match)
            torch.cuda.synchronize(dev)
            extra_lanes.append((c_x, s_x, a_x, m_x))
            lanes.append((c_x, s_x.cuda_stream, a_x, m_x, [a_x[c].data_ptr() for c in cams]))
    tick = [0]

    def step_h():
        """one step of the headline loop"""
        if not pipelined:
            return step()
        c_, s_, a_, m_, dp_ = lanes[tick[0] % len(lanes)]
        tick[0] += 1
        c_.describe_batch_dev(img_ptrs, W, H, W, kp_ptrs, [NKP] * len(cams), dp_, s_)
        c_.match_jobs_dev(a_.data_ptr(), abi_jobs, m_.data_ptr(), s_)

    def fence():
        torch.cuda.synchronize()            # every stream of the device
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # cold leg: W warm-up steps, then 20 timed steps, BEFORE any clock settling -- what a driver that only does `--warmup W` would see
    # (reported as value_cold / ms_per_step_cold; every rank runs the same steps)
    cold_steps = 20
    for _ in range(args.warmup):
        step_h()
    fence()
    tc = time.perf_counter()
    for _ in range(cold_steps):
        step_h()
    fence()
    dt_cold = time.perf_counter() - tc
    # clock settling (not part of W or K: the same step, untimed; every rank runs the same count -- a step may hold a collective)
    for i in range(max(args.settle_steps, 0)):
        step_h()
        if (i & 255) == 255:
            torch.cuda.synchronize()           # keep the queue bounded
    fence()
    for _ in range(args.warmup):
        step_h()
    fence()
    graph = None
    if args.graph and world == 1:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            step()
        graph.replay()
        fence()
    # timed region: K steps; the sweep kernel is bracketed by HIP events on its stream (both lanes' when the loop is pipelined)
    ctx.profile_reset()
    for lx in extra_lanes:
        lx[0].profile_reset()
    if graph is None:
        ctx.profile_enable(True, only=["k2nn_sweep_kernel"])
        for lx in extra_lanes:
            lx[0].profile_enable(True, only=["k2nn_sweep_kernel"])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            step_h()
    fence()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    one_stream = None
    pipelined_info = None
    if pipelined:
        pa = list(prof["k2nn_sweep_kernel"])
        same_lanes = True
        for lx in extra_lanes:
            lx[0].profile_enable(False)
            px = lx[0].profile_read()["k2nn_sweep_kernel"]
            pa[0] += px[0]; pa[1] += px[1]
            same_lanes = same_lanes and bool(torch.equal(d_match[:n_out], lx[3][:n_out]))
        pipelined_info = {"sweep_us_while_overlapped": pa[0] / max(pa[1], 1) * 1e3, "sweep_launches": pa[1],
                          "lanes_identical_results": same_lanes, "lanes": len(lanes)}
        # the same K steps on ONE stream (the device drained between a step's kernels): rounds 1-4's headline, and the region the
        # roofline's per-kernel duration comes from -- a kernel's duration says something about the kernel only while it has the machine
        want_h = d_match[:n_out].clone()
        # (its own settling: behind the heavier pipelined load the chip holds a lower clock for some tens of milliseconds -- a one-stream
        # region started straight behind it read 0.114 ms per step and a 28.9 us sweep where the same loop settled reads 0.099 / 25.3)
        for i in range(max(args.settle_steps // 2, args.warmup, 10)):
            step()
            if (i & 255) == 255:
                torch.cuda.synchronize()
        fence()
        for _ in range(args.warmup):
            step()
        fence()
        ctx.profile_reset()
        ctx.profile_enable(True, only=["k2nn_sweep_kernel"])
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_one = time.perf_counter() - t1
        ctx.profile_enable(False)
        prof = ctx.profile_read()
        one_stream = {"ms_per_step": dt_one / args.steps * 1e3, "value": total_cmp / (dt_one / args.steps) / 1e6, "steps": args.steps,
                      "identical_results": bool(torch.equal(want_h, d_match[:n_out])) and same_lanes,
                      "settle_steps": max(args.settle_steps // 2, args.warmup, 10),
                      "what": "the same K steps on ONE stream: every kernel of a step has the machine to itself (rounds 1-4's headline loop); "
                              "`roofline` is measured here"}
    # sustained leg: the same step back to back for >= sustain_seconds (no events inside), then the in-kernel clock of a
    # stamped diagnostic sweep launched straight behind it (MI355X guide, DVFS item 6)
    sustained = None
    clock = None
    if args.sustain_seconds > 0:
        # back-to-back steps run ~20 % faster than the event-bracketed burst the estimate comes from (no launch gaps): 1.3x margin
        n_sus = max(args.steps, int(1.3 * args.sustain_seconds / max(dt / args.steps, 1e-6)) + 1)
        if world > 1:
            # every rank must run the SAME number of steps (each step holds a collective): take rank 0's count
            t_n = torch.tensor([n_sus], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
            dist.broadcast(t_n, src=0)
            n_sus = int(t_n.item())
        fence()
        ts = time.perf_counter()
        for _ in range(n_sus):
            if graph is not None:
                graph.replay()
            else:
                step_h()
        if pipelined:
            torch.cuda.synchronize()           # the clock check below wants the sweep alone on the device, straight behind the load
        if formulation == "matrix" and abi_jobs:
            j0 = abi_jobs[0]
            try:
                clock = ctx.k2nn_clock_check(arena.data_ptr() + 64 * j0[0], j0[1], arena.data_ptr() + 64 * j0[2], j0[3],
                                             d_match.data_ptr(), sptr)
            except Exception as exc:          # diagnostic only
                clock = repr(exc)
        fence()
        dsus = time.perf_counter() - ts
        sustained = {"seconds": dsus, "steps": n_sus, "ms_per_step": dsus / n_sus * 1e3}
    # stage breakdown: a separate short pass with every kernel bracketed (not part of `value`)
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(min(args.steps, 20)):
        step()
    fence()
    ctx.profile_enable(False)
    prof_all = ctx.profile_read()
    if graph is not None:
        prof = prof_all

    # the exchange step on its own (every rank takes part; outside the timed region): SURVEY.md 8(e) asks for it separately
    allgather_us = None
    if world > 1 and args.backend == "nccl":
        fence()
        tg = time.perf_counter()
        for _ in range(20):
            dist.all_gather_into_tensor(arena.view(-1), mine.view(-1))
        torch.cuda.synchronize()
        allgather_us = (time.perf_counter() - tg) / 20 * 1e6
    tmax = torch.tensor([dt, sustained["seconds"] if sustained else 0.0, dt_cold], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0].item())
    dt_cold = float(tmax[2].item())
    if sustained:
        sustained["seconds"] = float(tmax[1].item())
        sustained["ms_per_step"] = sustained["seconds"] / sustained["steps"] * 1e3
    ms_per_step = dt / args.steps * 1e3
    errors = []
    warnings = []

    # ---- N > 1: the product's own exchange entry points beside the headline's (informational, guarded) ------------------
    # Every rank walks the same sequence; after each phase that can fail on ONE rank (communicator, IPC mapping, the steps) the
    # ranks agree on the outcome through torch.distributed before anyone enters the next collective, and a watchdog ends the
    # process (rank 0 prints the line first) if a leg does not come back.
    exchange_legs = None
    legs_aborted = False
    if world > 1 and not args.no_exchange_legs:
        # what rank 0 prints if a leg hangs: the contract's fields of the headline measurement, already complete at this point
        fallback_line = {"metric": "Mmatches/s (512-bit Hamming comparisons) at 10k kp/img, describe+match step",
                         "value": total_cmp / (dt / args.steps) / 1e6, "unit": "Mmatches/s", "n_gpus": world, "steps": args.steps,
                         "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                         "dtype": "fp4 (+-1) x fp4 -> f32 accumulate, exact integers (matrix pipe); fp32 sample coords", "data": "synthetic",
                         "config": {"workload": "config[3]: %d cameras one-per-GPU, 640x480 x 10k kp, all-gather + %d pairs"
                                                % (world, len(multicam.exhaustive_pairs(world))), "keypoints_per_image": NKP},
                         "roofline": None, "cpu_baseline": None}
        try:
            exchange_legs = run_exchange_legs(args, ctx, dist, torch, dev, world, rank, step, fence, d_match, n_out, mine, arena, counts,
                                              img_ptrs, kp_ptrs, sptr, mc, fallback_line)
        except Exception as exc:           # outside the legs' own guards: this rank's view of the group can no longer be trusted
            exchange_legs = {"error": "exchange legs aborted on rank %d: %r" % (rank, exc)}
            legs_aborted = True
        for k, v in exchange_legs.items():
            # matches that DIFFER are a correctness failure (non-zero exit after the line); a leg that could not run (no librccl, no IPC
            # mapping between these devices ...) is recorded in the line -- `collective_fallback`, the leg's `error` -- and is not
            warn = (isinstance(v, dict) and v.get("error")) or k == "error"
            if isinstance(v, dict) and v.get("identical") is False and not v.get("error"):
                errors.append("exchange_legs." + k)
            elif warn:
                warnings.append("exchange_legs." + k)

    # ---- promotion: the product's exchange becomes the headline when it ran, agreed with the torch step on every rank and was timed ---
    torch_exchange, collective_fallback, promoted = None, None, None
    if world > 1 and args.exchange == "auto" and mc is None:
        # --overlap-steps (default off until a multi-GPU node has measured it): the overlapped form of the product's exchange is the
        # candidate instead -- under the same conditions (ran, agreed with the torch step on every rank, was timed)
        leg_name = "clc-rccl-overlap" if args.overlap_steps else "clc-rccl"
        leg = (exchange_legs or {}).get(leg_name) if isinstance(exchange_legs, dict) else None
        if args.no_exchange_legs:
            collective_fallback = "--no-exchange-legs: the product's exchange was not attempted"
        elif not isinstance(leg, dict):
            collective_fallback = "the exchange legs did not run: %r" % (exchange_legs,)
        elif leg.get("error"):
            collective_fallback = leg_name + ": " + str(leg["error"])
        elif leg.get("identical") is not True:
            collective_fallback = leg_name + ": matches differ from the torch exchange's"
        elif not leg.get("seconds_max_over_ranks"):
            collective_fallback = leg_name + ": no timed region"
        else:
            promoted = leg
            torch_exchange = {"collective": "RCCL all_gather_into_tensor (torch.distributed)" if args.backend == "nccl" else "REHEARSAL: gloo all_gather staged through host memory",
                              "ms_per_step": ms_per_step, "value": total_cmp / (dt / args.steps) / 1e6, "steps": args.steps,
                              "what": "the same K timed steps with torch.distributed's all-gather as the exchange, measured first; the headline "
                                      "(value, ms_per_step, roofline) is the product's own exchange, checked against this one's matches"}
            dt = float(leg["seconds_max_over_ranks"])
            ms_per_step = dt / args.steps * 1e3
            if leg.get("sweep_prof_rank0"):
                prof = dict(prof)
                prof["k2nn_sweep_kernel"] = tuple(leg["sweep_prof_rank0"])

    if rank == 0:
        def avg_us(name, src=None):
            ms, cnt = (src or prof_all)[name]
            return (ms / cnt * 1e3) if cnt else None

        sweep_us = avg_us("k2nn_sweep_kernel", prof)
        clatch_us = avg_us("clatch_kernel")
        launches = prof["k2nn_sweep_kernel"][1]
        cmp_per_launch = my_cmp * (args.steps if graph is None else min(args.steps, 20)) / max(launches, 1)
        compulsory_bytes = sum(64 * (j.nq + counts[j.pair[1]]) + 4 * j.nq for j in jobs)
        roof = None
        if sweep_us:
            t = sweep_us * 1e-6
            # measured HBM traffic of this kernel comes from separate rocprofv3 --pmc passes (profiles/); it is attached
            # only when the recorded plan (formulation, shape) is the one that just ran, and says where it came from
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, "profiles", "r06_k2nn_hbm_traffic.json")
            if os.path.exists(tpath) and world == 1:
                try:
                    rec = json.load(open(tpath))
                    if rec.get("formulation") == formulation and rec.get("nq") == NKP and rec.get("nt") == NKP \
                            and not os.environ.get("CLC_K2NN_TARGET_BLOCKS") and not os.environ.get("CLC_K2NN_XCD_MAP"):
                        traffic = rec.get("hbm_bytes_per_launch")
                        traffic_source = "static: %s (%s)" % (os.path.relpath(tpath, ROOT), rec.get("how", "rocprofv3 --pmc"))
                except Exception:
                    traffic = None
            hbm = {"achieved_GBps": compulsory_bytes / t / 1e9, "peak_GBps": HBM_PEAK_GBS, "frac": compulsory_bytes / t / 1e9 / HBM_PEAK_GBS,
                   "algorithmic_bytes_per_launch": compulsory_bytes, "swept_GBps": cmp_per_launch * 64 / t / 1e9,
                   "note": "compulsory bytes 64 (nq + nt) + 4 nq per pair: the sweep is not HBM-bound"}
            if formulation == "matrix":
                flops = cmp_per_launch * 512 * 2
                roof = {"bound": "mfma", "kernel": "k2nn_sweep_mx_kernel", "achieved": flops / t / 1e12, "peak": MFMA_FP4_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": flops / t / 1e12 / MFMA_FP4_PEAK_TFLOPS, "traffic": traffic,
                        "traffic_source": traffic_source, "avg_launch_us": sweep_us,
                        "algorithmic_flop_per_launch": flops,
                        "peak_note": "dense FP4 (v_mfma_scale_f32_32x32x64_f8f6f4): 256 CU x 4 SIMD x 4096 flop/clk x 2.4 GHz; "
                                     "1 comparison = 512 exact +-1 multiply-adds (SURVEY 8d counts it as 32 VALU lane-ops)",
                        "Gcmp_per_s_kernel": cmp_per_launch / t / 1e9, "hbm": hbm,
                        # SURVEY.md 8(d) as written: 32 VALU lane-ops per comparison against 39.3 T lane-ops/s (the popcount formulation's
                        # bound).  Above 1 for this kernel because the work left the vector ALU for the matrix pipe: kept so that the
                        # line can be checked against the survey's own definition; `frac` above is the binding one.
                        "valu_8d": {"lane_ops_per_comparison": 32, "achieved_Tlaneops": cmp_per_launch * 32 / t / 1e12,
                                    "peak_Tlaneops": VALU_PEAK_TLANEOPS, "frac": cmp_per_launch * 32 / t / 1e12 / VALU_PEAK_TLANEOPS,
                                    "note": "exceeds 1: the comparisons run as FP4 multiply-adds on the matrix pipe, not as xor + popcount lane-ops"}}
            else:
                laneops = cmp_per_launch * 32 / t / 1e12
                roof = {"bound": "valu", "kernel": "k2nn_sweep_kernel", "achieved": laneops, "peak": VALU_PEAK_TLANEOPS,
                        "unit": "Tlaneop/s", "frac": laneops / VALU_PEAK_TLANEOPS, "traffic": traffic, "traffic_source": traffic_source,
                        "avg_launch_us": sweep_us,
                        "peak_note": "256 CU x 64 lanes/clk x 2.4 GHz: measured 4-cycle issue of v_bcnt / SGPR-operand ops",
                        "lane_ops_per_comparison": 32, "Gcmp_per_s_kernel": cmp_per_launch / t / 1e9, "hbm": hbm}
            if pipelined and roof is not None:
                roof["measured_in"] = ("the one-stream region of this run (K steps, the sweep bracketed by HIP events on its stream, the kernel alone on the "
                                       "device); in the headline region the sweep shares the device with the next step's describe launch: "
                                       "pipelined.sweep_us_while_overlapped")
            if isinstance(clock, tuple):
                roof["clock_ghz_in_kernel"] = {"median": clock[0], "min": clock[1], "max": clock[2], "workgroups": clock[3],
                                               "what": "s_memtime / s_memrealtime inside a stamped diagnostic sweep launched right behind the sustained leg",
                                               "frac_at_measured_clock": roof["frac"] * CLOCK_GHZ / clock[0] if formulation == "matrix" else None}
            elif clock is not None:
                roof["clock_ghz_in_kernel"] = {"error": clock}
        n_desc_launch = NKP * (1 if args.per_camera_launches else len(cams))
        out = {
            "metric": "Mmatches/s (512-bit Hamming comparisons) at 10k kp/img, describe+match step",
            "value": total_cmp / (dt / args.steps) / 1e6,
            "unit": "Mmatches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle": {"steps": max(args.settle_steps, 0), "what": "untimed steps before the warm-up so that the timed region runs at settled clocks"},
            "warmup_effective": args.warmup + cold_steps + max(args.settle_steps, 0) + args.warmup,
            "value_cold": total_cmp / (dt_cold / cold_steps) / 1e6, "ms_per_step_cold": dt_cold / cold_steps * 1e3,
            "cold_what": "%d timed steps straight behind the first %d warm-up steps of the process, before the settle steps" % (cold_steps, args.warmup),
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("fp4 (+-1) x fp4 -> f32 accumulate, exact integers (matrix pipe)" if formulation == "matrix" else "u32 (xor+popcount)")
                     + "; fp32 sample coords",
            "data": "synthetic",
            "config": {"workload": ("config[1]: 2 images 640x480 x 10k kp, CLATCH + K2NN 1 pair, thr 40" if world == 1 else
                                    "config[3]: %d cameras one-per-GPU, 640x480 x 10k kp, all-gather + %d pairs" % (world, len(multicam.exhaustive_pairs(world)))),
                       "keypoints_per_image": NKP, "pairs": len(multicam.exhaustive_pairs(n_cams)),
                       "comparisons_per_step": total_cmp, "k2nn_formulation": formulation,
                       "scene": "one rectangle scene, per-camera sensor noise sigma 2, same keypoints in camera-specific order"},
            "sustained": sustained,
            "sustained_value": (total_cmp / (sustained["ms_per_step"] * 1e-3) / 1e6) if sustained else None,
            "stages": {"clatch_us_per_launch": clatch_us, "pyramid_us_per_launch": avg_us("pyramid_kernel"),
                       "k2nn_sweep_us": sweep_us, "k2nn_merge_us": avg_us("k2nn_merge_kernel"),
                       "cameras_per_describe_launch": 1 if args.per_camera_launches else len(cams),
                       "Mdesc_per_s_kernel": (n_desc_launch / clatch_us) if clatch_us else None,
                       "Mdesc_per_s_step": len(cams) * NKP * world / (dt / args.steps) / 1e6},
            "roofline": roof,
            "k2nn_device": ctx.k2nn_device_info(),        # XCDs / CUs the sweep planner reads from the device, and where its unequal shares come from
            "launch_mode": ("hipGraph replay" if graph is not None else
                            ("eager launches, consecutive steps dealt to %d LANES in turn (a lane = context + stream + descriptor arena): step i's sweep runs "
                             "beside the next steps' pyramid + CLATCH; every step does all of its work, results identical to the one-stream loop's; "
                             "`one_stream` = the same K steps with the device drained between a step's kernels (rounds 1-4's value)" % len(lanes)) if pipelined
                            else "eager launches, one stream"),
            "one_stream": one_stream,
            "pipelined": pipelined_info,
            "collective": ("none" if world == 1 else
                           ("clc_mc_gather_enqueue_dev: " + promoted["exchange"] + " (product exchange, promoted after agreeing with the torch step)") if promoted else
                           (("RCCL all_gather_into_tensor" if mc is None else
                             "clc_mc_gather_enqueue_dev: " + ("ncclAllGather" if mc_mode == 0 else "IPC peer copies + 4-byte fence all-gather"))
                            if args.backend == "nccl" else "REHEARSAL: gloo all_gather staged through host memory")),
            "collective_fallback": collective_fallback,
            "torch_exchange": torch_exchange,
            "allgather_us_rank0": allgather_us,
            "exchange_legs": exchange_legs,
        }
        if world > 1:
            # what an efficiency computed from this line against the N = 1 line can and cannot show (SURVEY.md 8e defines E(G) against
            # G x the one-pair rate; the work per step is pairs x 10^8 comparisons and pairs = N (N - 1) / 2)
            pairs_n = len(multicam.exhaustive_pairs(world))
            out["efficiency_note"] = (
                "comparisons per step: N=1 holds 1 pair (2 cameras on one GPU) = %.0e; this line holds %d pair(s) = %.0e on %d GPUs. "
                "value(N)/(N x value(1)) is bounded by pairs/N x (N=1 step time / this step's critical path): at N=2 the work is the SAME "
                "1e8 comparisons as N=1 split over two GPUs (one camera's describe + half a pair each, + the exchange), so E(2) <= ~0.8 "
                "by construction (one-GPU projection without exchange: 61 us per step against 2 x 99.5 / 2); from N=4 on the pair count "
                "(6, 28) outgrows the rank count and E can exceed 1 against the one-pair N=1 line. cold_/sustained figures of this line were "
                "measured with the torch exchange." % (float(NKP) * NKP, pairs_n, float(pairs_n) * NKP * NKP, world))
        # Everything below is reported next to the headline line and never takes it down: each section runs guarded, a
        # failure is recorded under its own key -- and makes the process exit non-zero AFTER the line is printed.
        only = set(k for k in (args.sections or "").split(",") if k)

        def guarded(key, fn):
            if only and key not in only:
                return
            try:
                fn()
            except Exception as exc:
                out[key] = {"error": repr(exc)}
                errors.append(key)

        def sec_clatch():
            # CLATCH roofline (the other big kernel of the step): 512 v_dot4-quadruples = 98 304 LDS byte reads per descriptor
            #  against the LDS read peak; VALU instruction count from the PMC passes in profiles/
            if not clatch_us:
                return
            lds_bytes = 98304.0 * n_desc_launch
            out["stages"]["clatch_roofline"] = {
                "bound": "lds+valu", "algorithmic_lds_bytes_per_descriptor": 98304,
                "achieved_TBps": lds_bytes / (clatch_us * 1e-6) / 1e12, "peak_TBps": 256 * 256 * CLOCK_GHZ * 1e9 / 1e12,
                "frac": lds_bytes / (clatch_us * 1e-6) / 1e12 / (256 * 256 * CLOCK_GHZ * 1e9 / 1e12),
                "peak_note": "ds_read_b64: 256 B/clk/CU x 256 CU x 2.4 GHz"}

        def sec_k2nn_ab():
            # the other formulation of the sweep on the same descriptors, same run: A/B evidence (not part of `value`)
            if world != 1 or not abi_jobs:
                return
            other = "popcount" if formulation == "matrix" else "matrix"
            want = d_match[:n_out].clone()
            ctx.set_k2nn_formulation(other)
            try:
                ctx.profile_reset()
                ctx.profile_enable(True, only=["k2nn_sweep_kernel"])
                for _ in range(20):
                    ctx.match_jobs_dev(arena.data_ptr(), multicam.jobs_to_abi(
                        multicam.shard_pairs(counts, world, rank, grain=ctx.k2nn_queries_per_block), counts, NKP, THR), d_match.data_ptr(), sptr)
                torch.cuda.synchronize()
                ctx.profile_enable(False)
                pf = ctx.profile_read()
                same = bool(torch.equal(want, d_match[:n_out]))
            finally:
                ctx.set_k2nn_formulation(formulation)
            out["stages"]["k2nn_other_formulation"] = {"formulation": other, "sweep_us": pf["k2nn_sweep_kernel"][0] / max(pf["k2nn_sweep_kernel"][1], 1) * 1e3,
                                                       "identical_results": same}
            if not same:
                raise RuntimeError("the two K2NN formulations disagree")

        def sec_two_streams():
            # Throughput with consecutive steps OVERLAPPED (informational, not `value`): step i on stream / context / arena i & 1, so the
            # K2NN sweep of one step (matrix pipe) runs beside the pyramid + CLATCH of the next (vector ALU + LDS).  Every step does all
            # of its work and the results are checked against the sequential run; what changes is that the GPU is not drained between
            # a step's sweep and the next step's describe -- how a streaming host would drive it.
            if world != 1 or not abi_jobs or args.sustain_seconds <= 0 or pipelined:   # --sustain-seconds 0 (the rocprofv3 passes): one stream only,
                return                                                          # so that per-kernel averages are of kernels that had the machine;
                                                                                # pipelined: the headline loop IS this loop
            ctx2 = Context(device=dev_index, width=W, height=H, maxkp=NKP)
            try:
                st2 = torch.cuda.Stream(device=dev)
                arena2 = torch.zeros_like(arena)
                match2 = torch.empty_like(d_match)
                torch.cuda.synchronize(dev)
                lanes = [(ctx, sptr, arena, d_match, [arena[c].data_ptr() for c in cams]),
                         (ctx2, st2.cuda_stream, arena2, match2, [arena2[c].data_ptr() for c in cams])]
                want = d_match[:n_out].clone()

                def go(i):
                    c_, s_, a_, m_, dp_ = lanes[i & 1]
                    c_.describe_batch_dev(img_ptrs, W, H, W, kp_ptrs, [NKP] * len(cams), dp_, s_)
                    c_.match_jobs_dev(a_.data_ptr(), abi_jobs, m_.data_ptr(), s_)
                for i in range(20):
                    go(i)
                torch.cuda.synchronize()
                n = max(200, int(0.5 / max(ms_per_step * 1e-3, 1e-6)))
                t0 = time.perf_counter()
                for i in range(n):
                    go(i)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
                same = bool(torch.equal(want, match2[:n_out])) and bool(torch.equal(want, d_match[:n_out]))
                out["stages"]["two_streams_overlapped"] = {
                    "what": "consecutive steps alternate between two streams / contexts / descriptor arenas; identical results per step",
                    "steps": n, "ms_per_step": dt2 / n * 1e3, "Mmatches_per_s": total_cmp / (dt2 / n) / 1e6, "identical_results": same}
                if not same:
                    raise RuntimeError("overlapped steps produced different matches")
            finally:
                ctx2.close()

        def sec_front_end():
            # GPU-resident front end on a real frame (informational): pyramid -> FAST-9/NMS/orientation ->
            # CLATCH with the keypoint count kept in device memory (no host round trip)
            def fe():
                ctx.pyramid_build_dev(imgs[0].data_ptr(), W, H, W, sptr)
                ctx.detect_dev(sptr)
                ctx.describe_detected_dev(None, sptr)
            for _ in range(200):              # the section starts on an idle device: bring the clocks back up, let the host get ahead
                fe()
            torch.cuda.synchronize()          # rank-0-only section: no collective here
            ctx.profile_reset()
            ctx.profile_enable(True)
            for _ in range(200):
                fe()
            torch.cuda.synchronize()
            ctx.profile_enable(False)
            pf = ctx.profile_read()
            _, n_found = ctx.detect(capacity=1)
            t1 = time.perf_counter()
            for _ in range(200):
                fe()
            torch.cuda.synchronize()
            fe_us = (time.perf_counter() - t1) / 200 * 1e6
            out["front_end"] = {"what": "640x480 synthetic frame, all on device: pyramid + FAST-9/NMS/angle (8 levels, two launches) + CLATCH; "
                                        "*_us = HIP events around each stage's launches, 200 frames back to back behind 200 untimed ones",
                                "keypoints": int(n_found),
                                "pyramid_us": pf["pyramid_kernel"][0] / max(pf["pyramid_kernel"][1], 1) * 1e3,
                                "detect_us": pf["detect_kernels"][0] / max(pf["detect_kernels"][1], 1) * 1e3,
                                "clatch_us": pf["clatch_kernel"][0] / max(pf["clatch_kernel"][1], 1) * 1e3,
                                "frame_us_no_events": fe_us}
            # the batched entry point on 4 and 8 copies of the camera set (one pyramid, two detector, one CLATCH launch per call)
            capk = NKP
            bk = [torch.zeros((capk, 20), dtype=torch.uint8, device=dev) for _ in range(8)]
            bc = [torch.zeros((2,), dtype=torch.int32, device=dev) for _ in range(8)]
            bd = [torch.zeros((capk, 64), dtype=torch.uint8, device=dev) for _ in range(8)]
            torch.cuda.synchronize(dev)
            for nb in (4, 8):
                ip = [imgs[i % len(imgs)].data_ptr() for i in range(nb)]
                args_ = (ip, W, H, W, [t.data_ptr() for t in bk[:nb]], [t.data_ptr() for t in bc[:nb]], [t.data_ptr() for t in bd[:nb]], sptr)
                for _ in range(20):
                    ctx.detect_batch_dev(*args_)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(100):
                    ctx.detect_batch_dev(*args_)
                torch.cuda.synchronize()
                out["front_end"]["batch_of_%d_us_per_camera" % nb] = (time.perf_counter() - t1) / 100 / nb * 1e6

        def sec_pose():
            # p50 pose-solve (BASELINE metric, config[2] sizes) on host buffers, with the reference's model selection:
            # a-contrario RANSAC (Localizer.hpp:82-93: error_max = +inf, 256 iterations) + refinement + covariance
            # (localizeImage end to end, what the reference prints as "PNP in ms", coloc.hpp:222-225)
            pose = {}
            for n_pts in (200, 1000, 5000):
                sc = synth.pnp_scene(n_pts, seed=4000 + n_pts)
                ta, tl, its = [], [], []
                reps = 210                                   # >= 200 timed solves per size after the 5 warm-up ones
                for it in range(reps):
                    t1 = time.perf_counter()
                    r = ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], max_iteration=256, seed=it + 1)
                    ta.append((time.perf_counter() - t1) * 1e3)
                    its.append(r["iterations"])
                    t1 = time.perf_counter()
                    r2 = ctx.pnp_acransac(sc["X"], sc["x"], sc["K"], max_iteration=256, seed=it + 1, refine=True)
                    tl.append((time.perf_counter() - t1) * 1e3)
                ta, tl = np.sort(ta[5:]), np.sort(tl[5:])
                fx = []
                for it in range(45):                            # the cheaper fixed-threshold rule, for comparison only
                    t1 = time.perf_counter()
                    ctx.pnp_localize(sc["X"], sc["x"], sc["K"], n_samples=256, seed=it + 1, thr2=16.0)
                    fx.append((time.perf_counter() - t1) * 1e3)
                # the same two calls straight through the C ABI with the caller's buffers allocated once, as a C++ host makes them
                # (the Python wrapper above allocates its outputs and converts every argument on each call: 10-15 us)
                import ctypes as C
                from coloc_amd.abi import _p
                Xc, xc_, Kc_ = (np.ascontiguousarray(sc[k], dtype=np.float64) for k in ("X", "x", "K"))
                Rt_, cov_ = np.zeros(12), np.zeros(36)
                mk_, inl_ = np.zeros(n_pts, dtype=np.uint8), np.zeros(n_pts, dtype=np.int32)
                ni_, its_, em_, nfa_, rm_ = C.c_int(), C.c_int(), C.c_double(), C.c_double(), C.c_double()
                pX, px, pK, pRt, pcov, pmk, pinl = _p(Xc), _p(xc_), _p(Kc_.reshape(9)), _p(Rt_), _p(cov_), _p(mk_), _p(inl_)
                ca, cl = [], []
                for it in range(reps):
                    t1 = time.perf_counter()
                    rc1 = ctx.lib.clc_pnp_acransac(ctx.h, pX, px, n_pts, pK, 256, it + 1, float("inf"), pRt, pmk, pinl, C.byref(ni_), C.byref(em_),
                                                   C.byref(nfa_), C.byref(its_))
                    ca.append((time.perf_counter() - t1) * 1e3)
                    t1 = time.perf_counter()
                    rc2 = ctx.lib.clc_pnp_localize_ac(ctx.h, pX, px, n_pts, pK, 256, it + 1, float("inf"), 16.0, pRt, pcov, pmk, pinl, C.byref(ni_),
                                                      C.byref(em_), C.byref(rm_))
                    cl.append((time.perf_counter() - t1) * 1e3)
                    assert rc1 == 0 and rc2 == 0
                assert ni_.value == len(r2["inliers"]) and np.array_equal(Rt_.reshape(3, 4), r2["Rt"])     # same call, same answer
                ca, cl = np.sort(ca[5:]), np.sort(cl[5:])
                pose["N%d" % n_pts] = {"acransac_p50_ms": float(ta[len(ta) // 2]), "acransac_p95_ms": float(ta[int(len(ta) * 0.95)]),
                                       "c_abi_acransac_p50_ms": float(ca[len(ca) // 2]), "c_abi_with_refine_p50_ms": float(cl[len(cl) // 2]),
                                       "c_abi_with_refine_p95_ms": float(cl[int(len(cl) * 0.95)]),
                                       "with_refine_p50_ms": float(tl[len(tl) // 2]), "with_refine_p95_ms": float(tl[int(len(tl) * 0.95)]),
                                       "solves": int(len(ta)), "iterations_median": float(np.median(its)), "inliers": int(len(r2["inliers"])),
                                       "precision_found_px": float(r2["error_max"]),
                                       "fixed_threshold_with_refine_p50_ms": float(np.median(fx[5:]))}
            out["pose_solve"] = {"rule": "a-contrario RANSAC (AC-RANSAC, NFA over sorted residuals, no threshold given), P3P minimal solver, "
                                         "max_iteration 256, then LM/Huber(16) refinement + 6x6 covariance; host buffers in/out; *_p50_ms through the Python "
                                         "wrapper (coloc_amd.Context.pnp_acransac), c_abi_* = clc_pnp_acransac / clc_pnp_localize_ac called directly",
                                 "fixed_threshold_note": "clc_pnp_localize: 256 samples scored against a given 4 px threshold -- NOT the reference's rule, kept for comparison",
                                 **pose}
            out["pose_solve_p50_ms"] = pose["N1000"]["with_refine_p50_ms"]
            out["pose_acransac_only_p50_ms"] = pose["N1000"]["acransac_p50_ms"]
            out["pose_solve_p50_ms_c_abi"] = pose["N1000"]["c_abi_with_refine_p50_ms"]

        def sec_pose_batch():
            # BASELINE config[2]'s "batched PnP/RANSAC pose": the 4 cameras' localisations (N = 1000 each, 30 % outliers, a-contrario P3P +
            # refinement + covariance) in ONE clc_pnp_localize_ac_batch call -- every solve on a light context of its own, one host thread
            # driving all the chains of short launches so that they interleave on the device -- against the same 4 solves one after the
            # other (clc_pnp_localize_ac).  Caller's buffers allocated once, the C ABI called directly (as a C++ host does); results compared.
            import ctypes as C
            from coloc_amd.abi import PoseJob
            res = {}
            for ncam in (4, 8):
                pcs = [Context(device=dev_index, detector=False, matcher=False) for _ in range(ncam)]
                try:
                    scs = [synth.pnp_scene(1000, seed=4000 + c) for c in range(ncam)]
                    keep, jobs = [], (PoseJob * ncam)()
                    for c, sc in enumerate(scs):
                        X, x, K = (np.ascontiguousarray(sc[k], dtype=np.float64) for k in ("X", "x", "K"))
                        K = K.reshape(9)
                        Rt, cov, mk, inl = np.zeros(12), np.zeros(36), np.zeros(1000, dtype=np.uint8), np.zeros(1000, dtype=np.int32)
                        keep.append((X, x, K, Rt, cov, mk, inl))
                        j = jobs[c]
                        j.X, j.x, j.K, j.n, j.max_iteration, j.precision, j.refine, j.huber_a = X.ctypes.data, x.ctypes.data, K.ctypes.data, 1000, 256, float("inf"), 1, 16.0
                        j.Rt, j.cov, j.inlier_mask, j.inliers = Rt.ctypes.data, cov.ctypes.data, mk.ctypes.data, inl.ctypes.data
                    hs = (C.c_void_p * ncam)(*[c_.h for c_ in pcs])
                    tb, ts = [], []
                    ni_, em_, rm_ = C.c_int(), C.c_double(), C.c_double()
                    same = True
                    for it in range(105):
                        for c in range(ncam):
                            jobs[c].seed = it + 1 + c
                        t1 = time.perf_counter()
                        rc = ctx.lib.clc_pnp_localize_ac_batch(hs, jobs, ncam)
                        tb.append((time.perf_counter() - t1) * 1e3)
                        assert rc == 0
                        got = [(keep[c][3].copy(), int(jobs[c].n_inliers)) for c in range(ncam)]
                        t1 = time.perf_counter()
                        for c in range(ncam):
                            X, x, K, Rt, cov, mk, inl = keep[c]
                            rc = ctx.lib.clc_pnp_localize_ac(ctx.h, X.ctypes.data, x.ctypes.data, 1000, K.ctypes.data, 256, it + 1 + c, float("inf"), 16.0,
                                                             Rt.ctypes.data, cov.ctypes.data, mk.ctypes.data, inl.ctypes.data, C.byref(ni_), C.byref(em_), C.byref(rm_))
                            assert rc == 0
                            same = same and ni_.value == got[c][1] and np.array_equal(Rt, got[c][0])
                        ts.append((time.perf_counter() - t1) * 1e3)
                    tb, ts = np.sort(tb[5:]), np.sort(ts[5:])
                    res["cameras_%d" % ncam] = {"batch_p50_ms": float(tb[len(tb) // 2]), "batch_p95_ms": float(tb[int(len(tb) * 0.95)]),
                                                "per_pose_p50_ms": float(tb[len(tb) // 2]) / ncam,
                                                "one_after_the_other_p50_ms": float(ts[len(ts) // 2]), "one_after_the_other_per_pose_p50_ms": float(ts[len(ts) // 2]) / ncam,
                                                "batches": int(len(tb)), "identical_results": bool(same)}
                finally:
                    for c_ in pcs:
                        c_.close()
            out["pose_batch"] = {"what": "config[2] 'batched PnP/RANSAC pose': clc_pnp_localize_ac_batch, one a-contrario P3P solve + LM refinement + covariance per "
                                         "camera (N = 1000, 30 % outliers, 256 iterations max), all cameras' chains of launches interleaved from one host "
                                         "thread; C ABI, caller's buffers allocated once; beside it the same solves one after the other", **res}
            out["pose_batch_per_pose_p50_ms"] = res["cameras_4"]["per_pose_p50_ms"]

        def sec_two_view():
            # two-view filter: a-contrario five-point RANSAC over 1000 correspondences, 30 % outliers
            rng2 = np.random.default_rng(11)
            Nc = 1000
            Xs = np.stack([rng2.uniform(-5, 5, Nc), rng2.uniform(-5, 5, Nc), rng2.uniform(4, 20, Nc)], 1)
            Kc = np.array([[1000.0, 0, 640], [0, 1000.0, 360], [0, 0, 1]])
            ang = 0.2
            Rc = np.array([[np.cos(ang), 0, -np.sin(ang)], [0, 1, 0], [np.sin(ang), 0, np.cos(ang)]])
            p1 = Xs @ Kc.T; p1 = p1[:, :2] / p1[:, 2:3]
            p2 = (Xs @ Rc.T + np.array([0.5, 0.1, 0.2])) @ Kc.T; p2 = p2[:, :2] / p2[:, 2:3] + rng2.normal(0, 0.5, (Nc, 2))
            oi = rng2.choice(Nc, 300, replace=False)
            p2[oi] = np.stack([rng2.uniform(0, 1280, 300), rng2.uniform(0, 720, 300)], 1)
            te, tf = [], []
            for it in range(35):
                t1 = time.perf_counter()
                r = ctx.essential_acransac(p1, p2, Kc, Kc, (1280, 720), max_iteration=256, seed=it + 1)
                te.append((time.perf_counter() - t1) * 1e3)
                t1 = time.perf_counter()
                ctx.essential_ransac(p1, p2, Kc, Kc, n_samples=256, seed=it + 1, thr2=4.0)
                tf.append((time.perf_counter() - t1) * 1e3)
            out["two_view"] = {"what": "clc_essential_acransac: a-contrario five-point RANSAC (RobustMatcher.hpp:161-171), 256 iterations max, 1000 correspondences; "
                                       "a round = two launches since round 5 (replay + samples + five-point solve ~44 us; nfa ~11 us)",
                               "p50_ms": float(np.median(te[5:])), "inliers": int(len(r["inliers"])), "iterations": int(r["iterations"]),
                               "fixed_threshold_p50_ms": float(np.median(tf[5:]))}
            # the other two models RobustMatcher can filter with (RobustMatcher.hpp:128-151 'F', :188-239 'H'; round 6): the same
            # correspondences under the seven-point model, and the same cameras looking at a plane under the four-point one.  Their
            # rounds are ONE launch each (the solve is a few microseconds on one thread of every slot workgroup), like the resection's.
            nrm = np.array([0.1, -0.05, 1.0]); nrm /= np.linalg.norm(nrm)
            rays = np.c_[p1, np.ones(Nc)] @ np.linalg.inv(Kc).T
            Xp = rays * (9.0 / (rays @ nrm))[:, None]
            q2 = (Xp @ Rc.T + np.array([0.5, 0.1, 0.2])) @ Kc.T; q2 = q2[:, :2] / q2[:, 2:3] + rng2.normal(0, 0.5, (Nc, 2))
            q2[oi] = p2[oi]
            for mdl, b2 in (("F", p2), ("H", q2)):
                tm = []
                for it in range(35):
                    t1 = time.perf_counter()
                    rm = ctx.two_view_acransac(mdl, p1, b2, (1280, 720), max_iteration=256, seed=it + 1)
                    tm.append((time.perf_counter() - t1) * 1e3)
                out["two_view"]["model_" + mdl] = {"what": "clc_two_view_acransac '%s' (%s), same sizes" % (mdl, "seven-point, distance to the epipolar line" if mdl == "F" else "four-point, transfer error; planar scene"),
                                                   "p50_ms": float(np.median(tm[5:])), "inliers": int(len(rm["inliers"])), "iterations": int(rm["iterations"]),
                                                   "threshold_px": float(rm["error_max"])}
            # the same filter for 4 / 8 camera pairs in ONE clc_essential_acransac_batch call (what a frame of the streaming loop asks for):
            # the pairs' rounds share their launches (lockstep, blockIdx.y = pair; CLC_ACR_LOCKSTEP=0: chains of their own, interleaved --
            # 8 pairs 1.08-1.21 ms against 0.80), results job by job those of the single calls
            from coloc_amd.abi import essential_acransac_batch
            for npair in (4, 8):
                tcs = [Context(device=dev_index, detector=False, matcher=False) for _ in range(npair)]
                try:
                    tb, same = [], True
                    for it in range(30):
                        probs = [(p1, p2, Kc, Kc, (1280, 720), it + 1 + k) for k in range(npair)]
                        t1 = time.perf_counter()
                        rb = essential_acransac_batch(tcs, probs)
                        tb.append((time.perf_counter() - t1) * 1e3)
                    one = ctx.essential_acransac(p1, p2, Kc, Kc, (1280, 720), max_iteration=256, seed=30 + npair - 1)
                    same = bool(np.array_equal(rb[-1]["inliers"], one["inliers"]) and np.array_equal(rb[-1]["E"], one["E"]))
                    out["two_view"]["batch_of_%d" % npair] = {"per_pair_p50_ms": float(np.median(tb[5:])) / npair, "batch_p50_ms": float(np.median(tb[5:])),
                                                               "identical_to_single_call": same,
                                                               "launches": os.environ.get("CLC_ACR_LOCKSTEP", "shared by the pairs (lockstep rounds)")}
                    if not same:
                        raise RuntimeError("batched two-view filter differs from the single call")
                finally:
                    for c_ in tcs:
                        c_.close()

        def sec_shares():
            # What the driver's N = 2 / 4 / 8 runs will depend on, measured HERE on one GPU (informational, no collective, no
            # efficiency claim): every rank's share of the all-pairs sweep of an N-camera world (10k descriptors per camera, the
            # planner bench.py --gpus N uses), one camera's describe, and BASELINE config[2]'s launch group (4 cameras, 6 pairs).
            if world != 1:
                return
            big = torch.empty((8, NKP, 64), dtype=torch.uint8, device=dev)
            for c in range(8):
                big[c].copy_(arena[c & 1])
            out_buf = torch.empty((8 * NKP,), dtype=torch.int32, device=dev)

            def timed(fn, reps=30):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / reps * 1e6

            one_cam = timed(lambda: ctx.describe_batch_dev(img_ptrs[:1], W, H, W, kp_ptrs[:1], [NKP], [mine.data_ptr()], sptr))
            shares = {"describe_one_camera_us": one_cam,
                      "what": "back-to-back launches on one stream, host clock / 30; per-rank sweep = clc_match_jobs_dev over that rank's jobs"}
            for n_w in (2, 4, 8):
                cnts = [NKP] * n_w
                per_rank = []
                for r in range(n_w):
                    jb = multicam.jobs_to_abi(multicam.shard_pairs(cnts, n_w, r, grain=ctx.k2nn_queries_per_block), cnts, NKP, THR)
                    per_rank.append(timed(lambda: ctx.match_jobs_dev(big.data_ptr(), jb, out_buf.data_ptr(), sptr)))
                pairs_n = len(multicam.exhaustive_pairs(n_w))
                shares["N%d" % n_w] = {"pairs": pairs_n, "pair_shares_per_rank": pairs_n / n_w,
                                       "rank_sweep_us": [round(x, 1) for x in per_rank], "rank_sweep_us_max": max(per_rank),
                                       "step_us_without_exchange": one_cam + max(per_rank),
                                       "Mmatches_per_s_without_exchange": pairs_n * NKP * NKP / (one_cam + max(per_rank))}
            out["stages"]["shares"] = shares
            # config[2]: 4 cameras on ONE GPU, all 6 pairs in one launch group (descriptors of 4 x 10k keypoints at 1280 x 720)
            W2, H2 = 1280, 720
            ctx2 = Context(device=dev_index, width=W2, height=H2, maxkp=NKP)
            try:
                imgs2 = [torch.from_numpy(np.ascontiguousarray(synth.rect_image(W2, H2, seed=1000, noise_sigma=2.0 + c))).to(dev) for c in range(4)]
                kp2 = [torch.from_numpy(synth.random_keypoints(NKP, W2, H2, seed=2200 + c).view(np.uint8).reshape(-1, 20).copy()).to(dev)
                       for c in range(4)]
                ar2 = torch.zeros((4, NKP, 64), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize(dev)
                cnts = [NKP] * 4
                jb = multicam.jobs_to_abi(multicam.shard_pairs(cnts, 1, 0, grain=ctx2.k2nn_queries_per_block), cnts, NKP, THR)
                ip, kp_ = [t.data_ptr() for t in imgs2], [t.data_ptr() for t in kp2]
                dp = [ar2[c].data_ptr() for c in range(4)]
                t_desc = timed(lambda: ctx2.describe_batch_dev(ip, W2, H2, W2, kp_, cnts, dp, sptr))
                t_match = timed(lambda: ctx2.match_jobs_dev(ar2.data_ptr(), jb, out_buf.data_ptr(), sptr))

                def both():
                    ctx2.describe_batch_dev(ip, W2, H2, W2, kp_, cnts, dp, sptr)
                    ctx2.match_jobs_dev(ar2.data_ptr(), jb, out_buf.data_ptr(), sptr)
                t_step = timed(both)
                out["stages"]["config2"] = {"what": "BASELINE config[2] on one GPU: 4 cameras 1280x720 x 10k kp, one pyramid + one CLATCH launch, "
                                                    "all 6 pairs in one sweep launch (pose: pose_solve)",
                                            "describe_4_cameras_us": t_desc, "match_6_pairs_us": t_match, "step_us": t_step,
                                            "Mmatches_per_s": 6 * NKP * NKP / t_step, "Mdesc_per_s": 4 * NKP / t_step}
                # the same step with the REAL front end: the four frames' keypoints come from the GPU detector (clc_detect_batch_dev: one
                # pyramid, two detector and one CLATCH launch), the pairs are matched with the counts read on the device
                kb = [torch.zeros((NKP, 20), dtype=torch.uint8, device=dev) for _ in range(4)]
                cb = [torch.zeros((2,), dtype=torch.int32, device=dev) for _ in range(4)]
                torch.cuda.synchronize(dev)
                kbp, cbp = [t.data_ptr() for t in kb], [t.data_ptr() for t in cb]
                prs = multicam.exhaustive_pairs(4)
                jb2 = [(a_ * NKP, NKP, b_ * NKP, NKP, k_ * NKP, THR) for k_, (a_, b_) in enumerate(prs)]
                out6 = torch.empty((6 * NKP,), dtype=torch.int32, device=dev)

                def real_step():
                    ctx2.detect_batch_dev(ip, W2, H2, W2, kbp, cbp, dp, sptr)
                    ctx2.match_jobs_counted_dev(ar2.data_ptr(), jb2, [cbp[a_] for a_, _ in prs], [cbp[b_] for _, b_ in prs], [0] * 6,
                                                out6.data_ptr(), sptr)
                t_front = timed(lambda: ctx2.detect_batch_dev(ip, W2, H2, W2, kbp, cbp, dp, sptr))
                t_real = timed(real_step)
                torch.cuda.synchronize()
                n_kp = [int(t[0].item()) for t in cb]
                cmp_real = sum(n_kp[a_] * n_kp[b_] for a_, b_ in prs)
                out["stages"]["config2"]["real_front_end"] = {
                    "what": "detect -> describe of 4 x 1280x720 rendered-rectangle frames in one batched call + the 6 pair sweeps on the "
                            "detected keypoints (counts read on the device)",
                    "keypoints": n_kp, "detect_describe_4_cameras_us": t_front, "step_us": t_real,
                    "Mmatches_per_s": cmp_real / t_real, "Mdesc_per_s": sum(n_kp) / t_real}
            finally:
                ctx2.close()

        def sec_host_path():
            if world == 1:
                # accepted matches, and the end-to-end rates of the host-buffer entry points (uploads, downloads and
                # the synchronisation included; never used as `value`)
                acc = int((d_match[:n_out] >= 0).sum().item())
                out["accepted_matches_per_step"] = acc
                out["accepted_matches_per_s"] = acc / (dt / args.steps)
                hq, ht = arena[0].cpu().numpy(), arena[1].cpu().numpy()
                hk = kps_np[0]
                himg = imgs[0].cpu().numpy()
                tm, td = [], []
                for it in range(25):
                    t1 = time.perf_counter(); ctx.match_2nn(hq, ht, THR); tm.append(time.perf_counter() - t1)
                    t1 = time.perf_counter(); ctx.pyramid_build(himg); ctx.describe(hk); td.append(time.perf_counter() - t1)
                tm, td = float(np.median(tm[5:])), float(np.median(td[5:]))
                # the same call when the detector has PUBLISHED the two blocks (clc_desc_cache_publish: what HIPDetector does for the
                # regions it fills): the match finds the rows on the device, only the 40 KB of indices cross PCIe
                want_m = ctx.match_2nn(hq.copy(), ht.copy(), THR)
                tcache = {}
                for mode in ("verify", "trust"):
                    ctx.desc_cache_mode(mode)
                    ctx.desc_cache_publish(hq, d_src=arena[0].data_ptr())
                    ctx.desc_cache_publish(ht, d_src=arena[1].data_ptr())
                    tc = []
                    for it in range(25):
                        t1 = time.perf_counter(); got_m = ctx.match_2nn(hq, ht, THR); tc.append(time.perf_counter() - t1)
                    tcache[mode] = float(np.median(tc[5:]))
                    if not np.array_equal(want_m, got_m):
                        raise RuntimeError("cached and uploaded descriptor blocks gave different matches (%s)" % mode)
                ctx.desc_cache_mode("verify")
                out["host_path"] = {"what": "same work through the host-pointer entry points (PCIe copies + one sync per call included)",
                                    "match_2nn_10k_x_10k_us": tm * 1e6, "Mmatches_per_s_incl_transfers": NKP * NKP / tm / 1e6,
                                    "match_2nn_10k_x_10k_published_blocks_us": tcache["trust"] * 1e6,
                                    "match_2nn_10k_x_10k_published_blocks_verified_us": tcache["verify"] * 1e6,
                                    "published_what": "both descriptor blocks published by the front end (clc_desc_cache_publish): no upload, the call is sweep + 40 KB "
                                                      "of indices back + one synchronisation; identical matches.  *_verified_us: the DEFAULT mode (policy classes "
                                                      "included) -- the sweep starts on the device rows at once and the host folds both 640 KB blocks WHILE it runs, "
                                                      "comparing with the folds taken at publish time (an edited block is uploaded and swept again); *_us: a TRUSTING "
                                                      "context (opt-in: address + count + generation + 18 sampled rows, no pass over the block)",
                                    "pyramid_plus_describe_10k_us": td * 1e6, "Mdesc_per_s_incl_transfers": NKP / td / 1e6}

        def sec_policy_path():
            # The DROP-IN path timed from C++: tests/host/bench_policy.cpp drives HIPDetector -> HIPMatcher ->
            # HIPLocalizer in ColoC::mainThread's order (reference include/coloc/coloc.hpp:111-148) on rendered frames of one scene and
            # prints p50 / p95 of the spans the reference prints (:129-136 detection, :161-164 pair matching, :217-221 map tracking,
            # :222-225 PnP).  Host buffers in, OpenMVG-shaped regions / matches / pose out, every copy and synchronisation of the policy
            # classes included.  Never `value`.
            import subprocess
            import tempfile
            from coloc_amd import Context, keypoints_to_features
            K = np.array([[520.0, 0, 320.0], [0, 520.0, 240.0], [0, 0, 1.0]])
            relief = synth.smooth_relief()
            Ra, ta = synth.look_at_plane_pose((7.0, 7.0), 5.0, yaw=0.0, tilt=(0.10, -0.06))
            Rb, tb = synth.look_at_plane_pose((7.6, 6.7), 5.2, yaw=0.12, tilt=(-0.08, 0.09))
            Rc, tc = synth.look_at_plane_pose((6.5, 7.3), 4.9, yaw=-0.10, tilt=(0.05, 0.07))
            det = Context(device=dev_index, width=W, height=H, maxkp=12000, matcher=False)
            try:
                frames = None
                for n_rect in (900, 2500, 6000):          # a texture dense enough for ~5 k keypoints at 640 x 480
                    tex = synth.plane_texture(n_rect=n_rect)
                    frames = [synth.render_plane(tex, 100.0, K, R, t, W, H, relief=relief) for R, t in ((Rc, tc), (Rb, tb), (Ra, ta))]
                    kps0, desc0, _ = det.detect_and_describe(frames[0])
                    if len(kps0) >= 4500:
                        break
                kps_map, _, _ = det.detect_and_describe(frames[2])         # the keyframe the map is made of
                # the same frame through the device-pointer path: what the policy path's regions must hold
                d_img = torch.from_numpy(frames[0]).to(dev)
                det.pyramid_build_dev(d_img.data_ptr(), W, H, W, None)
                kp_dev, _ = det.detect(capacity=12000)
                tmp = torch.zeros((max(len(kp_dev), 1), 64), dtype=torch.uint8, device=dev)
                torch.cuda.synchronize(dev)
                det.describe_detected_dev(tmp.data_ptr(), None)
                det.sync()
                desc_dev = tmp.cpu().numpy()[:len(kp_dev)]
                # the same frame's front end device-resident, back to back (what front_end.frame_us_no_events is for bench's own frame)
                def fe_dev():
                    det.pyramid_build_dev(d_img.data_ptr(), W, H, W, None)
                    det.detect_dev(None)
                    det.describe_detected_dev(tmp.data_ptr(), None)
                for _ in range(200):
                    fe_dev()
                det.sync()
                t1 = time.perf_counter()
                for _ in range(200):
                    fe_dev()
                det.sync()
                fe_dev_us = (time.perf_counter() - t1) / 200 * 1e6
            finally:
                det.close()
            with tempfile.TemporaryDirectory() as d:
                for name, img in zip(("cam0", "cam1", "cam_map"), frames):
                    with open(os.path.join(d, name + ".pgm"), "wb") as f:
                        f.write(b"P5\n# rendered\n%d %d\n255\n" % (W, H))
                        f.write(img.tobytes())
                feat0 = keypoints_to_features(kps_map)
                synth.backproject_to_plane(feat0[:, :2].astype(np.float64), K, Ra, ta, relief=relief).astype(np.float64).tofile(os.path.join(d, "map_xyz.bin"))
                exe = os.path.join(d, "bench_policy")
                libdir = os.path.join(ROOT, "coloc_amd", "lib")
                subprocess.check_call(["g++", "-std=c++14", "-O2", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "coloc_amd", "host"),
                                       os.path.join(ROOT, "tests", "host", "bench_policy.cpp"), "-o", exe, "-L", libdir, "-lcoloc_hip", "-Wl,-rpath," + libdir])
                res = {}
                for name, env_publish in (("default", None), ("upload_every_call", "0"), ("trusting", "t")):
                    env = dict(os.environ)
                    if env_publish is not None:
                        env["BENCH_POLICY_PUBLISH"] = env_publish
                    r = subprocess.run([exe, d, str(W), str(H), str(K[0, 0]), str(K[0, 2]), str(K[1, 2]), "300", "60", "12000"], capture_output=True, text=True,
                                       cwd=d, env=env, timeout=240)
                    line = [l for l in r.stdout.splitlines() if l.startswith("POLICY ")]
                    if r.returncode != 0 or not line:
                        raise RuntimeError("bench_policy (%s) failed: rc %d %s" % (name, r.returncode, r.stderr[-500:]))
                    res[name] = json.loads(line[0][7:])
                got = np.fromfile(os.path.join(d, "policy_desc0.bin"), dtype=np.uint8).reshape(-1, 64)
                same = got.shape == desc_dev.shape and bool(np.array_equal(got, desc_dev)) and got.shape == desc0.shape and bool(np.array_equal(got, desc0))
            if not same:
                raise RuntimeError("the policy path's descriptors differ from the device-pointer path's")
            r0 = res["default"]
            out["policy_path"] = {
                "what": "tests/host/bench_policy.cpp: HIPDetector::detectFeaturesImage -> HIPMatcher::matchSceneWithMap -> HIPLocalizer::localizeImage per camera "
                        "frame (ColoC::mainThread's order, coloc.hpp:111-148), p50 over 300 frames behind 60 untimed ones, host buffers in / regions, matches, "
                        "pose + covariance out; pair_match_us = HIPMatcher::computeMatches of the two cameras (initMap, :161-164)",
                "frame": "%dx%d rendered scene" % (W, H), "keypoints": r0["keypoints"], "map_points": r0["map_points"], "map_matches": r0["map_matches"],
                "pose_inliers": r0["pose_inliers"],
                "detect_us": r0["detect_us"], "detect_steps_us": r0.get("detect_steps_us"), "match_us": r0["match_us"], "pose_us": r0["pose_us"], "frame_us": r0["frame_us"], "pair_match_us": r0["pair_match_us"],
                "p95": {k: r0[k + "_p95"] for k in ("detect_us", "match_us", "pose_us", "frame_us", "pair_match_us")},
                "device_front_end_us_same_frame": fe_dev_us,
                "pcie_bytes_per_frame": {"in": W * H, "out": int(r0["keypoints"][1]) * 84 + 8},
                "identical_to_device_pointer_path": same, "same_results_every_frame": r0["same_results_every_frame"],
                "descriptor_hand_over": {"default (published, verified behind the sweep)": {k: r0[k] for k in ("match_us", "pair_match_us")},
                                         "upload every call (reference behaviour)": {k: res["upload_every_call"][k] for k in ("match_us", "pair_match_us")},
                                         "trusting (opt-in)": {k: res["trusting"][k] for k in ("match_us", "pair_match_us")}},
            }

        def sec_cpu_baseline():
            if not args.no_cpu_baseline and world == 1:
                dq = arena[0].cpu().numpy()
                dt_ = arena[1].cpu().numpy()
                xy = [np.stack([k["x"], k["y"]], axis=1).astype(np.float32) for k in kps_np[:2]]
                out["cpu_baseline"] = cpu_baseline(dq, dt_, xy[0], xy[1])
                out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
                out["gpu_over_cpu_openmvg_ratio_rule"] = out["value"] / out["cpu_baseline"]["openmvg_ratio_rule"]["value"]
                if "best_effort_simd" in out["cpu_baseline"]:
                    out["gpu_over_cpu_best_effort_simd"] = out["value"] / out["cpu_baseline"]["best_effort_simd"]["value"]

        guarded("clatch_roofline", sec_clatch)
        if world == 1 and not args.headline_only:
            # the side sections belong to the one-GPU line; at N > 1 the other ranks would sit in the teardown barrier below while
            # rank 0 alone ran them for tens of seconds
            guarded("host_path", sec_host_path)
            guarded("policy_path", sec_policy_path)
            guarded("k2nn_ab", sec_k2nn_ab)
            guarded("two_streams", sec_two_streams)
            guarded("shares", sec_shares)
            guarded("front_end", sec_front_end)
            guarded("pose_solve", sec_pose)
            guarded("pose_batch", sec_pose_batch)
            guarded("two_view", sec_two_view)
            guarded("cpu_baseline", sec_cpu_baseline)
        if world > 1:
            # the side sections belong to the one-GPU line (cpu_baseline: rank 0 at N = 1 only, by the bench contract); their keys are
            # present and null here so that lines of different N can be compared field by field
            for k in ("cpu_baseline", "gpu_over_cpu", "front_end", "pose_solve", "pose_solve_p50_ms", "pose_batch", "pose_batch_per_pose_p50_ms", "two_view", "host_path", "policy_path",
                      "accepted_matches_per_step"):
                out.setdefault(k, None)
        if errors:
            out["section_errors"] = errors
        if warnings:
            out["section_warnings"] = warnings
        print(json.dumps(out), flush=True)
    if legs_aborted:                    # no further collective with a group in an unknown state: the line is out, leave
        sys.stdout.flush()
        os._exit(3)
    if mc is not None:
        mc.close()
    for lx in extra_lanes:
        lx[0].close()
    ctx.close()
    if world > 1:
        dist.barrier()                  # nobody tears the group down while another rank still uses it
        dist.destroy_process_group()
    if errors:
        sys.exit(3)


if __name__ == "__main__":
    main()
