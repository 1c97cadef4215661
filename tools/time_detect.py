"""Front-end timing on a real frame (run on the GPU box): pyramid -> detect (two launches) -> CLATCH, one camera and batches.
python3 tools/time_detect.py [reps]   -- under rocprofv3 --kernel-trace --stats for the per-kernel durations."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
for (W, H, batches) in ((640, 480, (1, 4, 8)), (1280, 720, (1, 4))):
    cap = 20000
    ctx = Context(device=0, width=W, height=H, maxkp=cap)
    imgs = [torch.from_numpy(synth.rect_image(W, H, seed=1000 + c, noise_sigma=2.0)).to(dev) for c in range(8)]
    kps = [torch.zeros((cap, 20), dtype=torch.uint8, device=dev) for _ in range(8)]
    cnt = [torch.zeros((2,), dtype=torch.int32, device=dev) for _ in range(8)]
    desc = [torch.zeros((cap, 64), dtype=torch.uint8, device=dev) for _ in range(8)]
    # single-camera entry points with per-kernel events
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(reps):
        ctx.pyramid_build_dev(imgs[0].data_ptr(), W, H, W, s); ctx.detect_dev(s); ctx.describe_detected_dev(None, s)
    torch.cuda.synchronize(); ctx.profile_enable(False)
    p = ctx.profile_read()
    _, found = ctx.detect(capacity=1)
    print("%dx%d single camera (%d kp): " % (W, H, found) + ", ".join("%s %.1f us" % (k, v[0] / max(v[1], 1) * 1e3) for k, v in p.items() if v[1]))
    for n in batches:
        ip, kp_, cp, dp = [t.data_ptr() for t in imgs[:n]], [t.data_ptr() for t in kps[:n]], [t.data_ptr() for t in cnt[:n]], [t.data_ptr() for t in desc[:n]]
        for with_desc in (False, True):
            for _ in range(20):
                ctx.detect_batch_dev(ip, W, H, W, kp_, cp, dp if with_desc else None, s)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.detect_batch_dev(ip, W, H, W, kp_, cp, dp if with_desc else None, s)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps * 1e6
            print("  batch of %d: pyramid + detect%s %.1f us per call (%.1f us per camera)" % (n, " + CLATCH" if with_desc else "", dt, dt / n))
    ctx.close()
