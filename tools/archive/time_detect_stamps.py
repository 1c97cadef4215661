"""[needs the one-launch detector of profiles/r05_detect_fused_emit.patch (apply it, then tools/build_variant.sh det_stamps -I../../include -DCLC_DET_STAMPS): the shipped library has no stamp symbol]
timeline of the detector launch's LAST workgroup (variant build with -DCLC_DET_STAMPS: COLOC_HIP_LIB=tools/bin/det_stamps.so);
s_memrealtime stamps (10 ns): entry, tile done, stores acknowledged, arrival returned, prefix known, band emitted, done counted, end."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
from coloc_amd import abi
W, H = 640, 480
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
ctx = Context(device=0, width=W, height=H, maxkp=20000)
img = torch.from_numpy(synth.rect_image(W, H, seed=1000, noise_sigma=2.0)).to(dev)
ctx.pyramid_build_dev(img.data_ptr(), W, H, W, s)
lib = ctypes.CDLL(os.environ["COLOC_HIP_LIB"])
out = (ctypes.c_ulonglong * 16)()
names = ["entry", "tile done", "stores acked", "arrived", "prefix", "emitted", "done counted", "end"]
for it in range(330):
    ctx.detect_dev(s)
    if it >= 320:
        torch.cuda.synchronize()
        lib.clc_dbg_det_stamps(out)
        t = [int(x) for x in out]
        print("tile %4d  block0 entry -> " % t[9] + "  ".join("%s %+.2f" % (n, (t[k] - t[8]) / 100.0) for k, n in enumerate(names)), "us")
ctx.close()
