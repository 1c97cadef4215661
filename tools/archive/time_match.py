#!/usr/bin/env python3
"""Device-side time of clc_match_2nn_dev for a few (nq, nt) shapes (HIP events on the launch stream, median of 50)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from coloc_amd import Context

def main():
    dev = torch.device("cuda", 0)
    ctx = Context(device=0, width=640, height=480, maxkp=20000)
    st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
    rng = np.random.default_rng(1)
    out = []
    for nq, nt in [(500, 5000), (5000, 500), (2000, 2000), (4000, 4000), (10000, 10000), (1000, 20000), (20000, 20000), (128, 128)]:
        Q = torch.from_numpy(rng.integers(0, 256, (nq, 64), dtype=np.uint8)).to(dev)
        T = torch.from_numpy(rng.integers(0, 256, (nt, 64), dtype=np.uint8)).to(dev)
        m = torch.empty(nq, dtype=torch.int32, device=dev)
        for _ in range(5):
            ctx.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream)
        ts = []
        for _ in range(50):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); ctx.match_2nn_dev(Q.data_ptr(), nq, T.data_ptr(), nt, 40, m.data_ptr(), st.cuda_stream); b.record(st)
            b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        ts.sort()
        out.append("%6d x %6d  %8.1f us  %7.1f Gcmp/s" % (nq, nt, ts[25], nq * nt / ts[25] / 1e3))
    print("\n".join(out))
    ctx.close()

if __name__ == "__main__":
    main()
