"""[needs a build with the stamp instrumentation of profiles/r05_detect_fused_emit.patch (apply it, then tools/build_variant.sh det_stamps -I../../include -DCLC_DET_STAMPS): the shipped library has no stamp symbol]
phase timeline of two workgroups of detect_tile_kernel (variant build -DCLC_DET_STAMPS, CLC_DETECT_LAUNCHES=2): block 0 (a tile of the
level that replays the KFAST.h:245 walk) and block 500; s_memrealtime stamps (10 ns) relative to the workgroup's entry."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
W, H = 640, 480
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
ctx = Context(device=0, width=W, height=H, maxkp=20000)
img = torch.from_numpy(synth.rect_image(W, H, seed=1000, noise_sigma=2.0)).to(dev)
ctx.pyramid_build_dev(img.data_ptr(), W, H, W, s)
lib = ctypes.CDLL(os.environ["COLOC_HIP_LIB"])
out = (ctypes.c_ulonglong * 16)()
names = ["staged", "pre-test", "score", "replay verdict", "suppressed"]
for it in range(330):
    ctx.detect_dev(s)
    if it >= 322:
        torch.cuda.synchronize()
        lib.clc_dbg_det_stamps(out)
        t = [int(x) for x in out]
        for o, nm in ((0, "block 0 (walk)"), (8, "block 500")):
            print("%-15s " % nm + "  ".join("%s %+.2f" % (n, (t[o + 1 + k] - t[o]) / 100.0) for k, n in enumerate(names)), "us;  entry vs block 0 %+.2f" % ((t[o] - t[0]) / 100.0))
ctx.close()
