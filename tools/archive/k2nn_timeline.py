"""Per-workgroup timeline of one 10k x 10k matrix-formulation sweep (stamped diagnostic build, CLC_K2NN_STAMP_DUMP):
when each workgroup enters, starts / ends its tile loop and has folded its results in, on the constant 100 MHz clock.
usage: k2nn_timeline.py   (run ON the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
ctx = Context(device=0, width=640, height=480, maxkp=70000, detector=False)
NQ, NT = (int(v) for v in os.environ.get("SIZE", "10000x10000").split("x"))
Q, T = synth.planted_descriptors(NQ, NT, seed=3000)
dq, dt = torch.from_numpy(Q).cuda(), torch.from_numpy(T).cuda()
dm = torch.empty(NQ, dtype=torch.int32, device="cuda")
for _ in range(200):
    ctx.match_2nn_dev(dq.data_ptr(), NQ, dt.data_ptr(), NT, 40, dm.data_ptr())
path = "/tmp/k2nn_stamps.bin"
os.environ["CLC_K2NN_STAMP_DUMP"] = path
clk = ctx.k2nn_clock_check(dq.data_ptr(), NQ, dt.data_ptr(), NT, dm.data_ptr())
print("plan", ctx.k2nn_plan_query(NQ, NT))
h = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
keep = h[:, 4] > 0
ids = np.nonzero(keep)[0]
h = h[keep]
t0 = h[:, 4].min()
ent, ls, le, ex = [(h[:, i].astype(np.int64) - int(t0)) * 0.01 for i in (4, 1, 3, 5)]       # microseconds
xcc = h[:, 7] & 0xF
cyc = (h[:, 2] - h[:, 0]).astype(np.float64)
print("clock", clk)
print("workgroups %d" % len(h))
q = lambda v: "min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(v, [0, 10, 50, 90, 100]))
print("entry            us:", q(ent))
print("loop start       us:", q(ls))
print("loop end         us:", q(le))
print("folded in        us:", q(ex))
print("prologue (entry -> loop start) us:", q(ls - ent))
print("loop length      us:", q(le - ls), " cycles:", q(cyc))
print("epilogue + fold  us:", q(ex - le))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCC %d: %3d workgroups, entry p50 %.2f, loop end p50 %.2f max %.2f, folded max %.2f" % (x, m.sum(), np.median(ent[m]), np.median(le[m]), le[m].max(), ex[m].max()))
# how many workgroups are inside their loop at each instant
for tt in np.arange(0, ex.max() + 1, 2.0):
    print("  t=%5.1f us: %3d entered, %3d in loop, %3d done" % (tt, (ent <= tt).sum(), ((ls <= tt) & (le > tt)).sum(), (ex <= tt).sum()))
# loop length against the dispatch order (linear workgroup id = stamp slot): is "oldest on the CU runs fastest" visible?
wg = ids                        # rows are in blockIdx.x order (one job)
for lo in range(0, len(h), 128):
    m = (wg >= lo) & (wg < lo + 128)
    print("  workgroups %3d-%3d: loop length p50 %.2f us, loop end p50 %.2f us" % (lo, min(lo + 127, len(h) - 1), np.median((le - ls)[m]), np.median(le[m])))

# where each workgroup sat (HW_ID of its wave 0): does the wave slot -- the order in which a CU's workgroups arrived -- decide who is served?
hw = (h[:, 7] >> 8).astype(np.int64)
slot, simd, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
print("wave slot of wave 0:", {int(v): int((slot == v).sum()) for v in np.unique(slot)})
for v in np.unique(slot):
    m = slot == v
    print("  slot %d: %3d workgroups, loop length p50 %.2f us (p10 %.2f, p90 %.2f), loop end p50 %.2f, p90 %.2f" % (v, m.sum(), np.median((le - ls)[m]), np.percentile((le - ls)[m], 10), np.percentile((le - ls)[m], 90), np.median(le[m]), np.percentile(le[m], 90)))
for name, f in (("id // 256", lambda v: v // 256),):
    print("  slot == %-14s for %4d of %d workgroups" % (name, int((f(ids) == slot).sum()), len(ids)))
