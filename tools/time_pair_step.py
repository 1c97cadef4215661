"""The pair step (clc_describe_match_pair_dev) against the two calls it replaces, bench.py's config[1] inputs (2 x 640x480 x 10k keypoints,
threshold 40): per configuration 400 settling steps, then 3 x 300 timed steps back to back on one stream (host clock / 300), alternated with
the baseline; matches compared with the baseline's.  Configurations: chunks = 1 (no overlap) / K equal chunks / CLC_PAIR_CHUNKS lists, each
with CLC_PAIR_TARGET_BLOCKS in (0 = 768, 512, 384, 256).  usage: time_pair_step.py [quick]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
W, H, N, THR = 640, 480, 10000, 40
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); s = st.cuda_stream
scene = synth.rect_image(W, H, seed=1000, noise_sigma=0.0).astype(np.float32)
base = synth.random_keypoints(N, W, H, seed=2000)
imgs = [torch.from_numpy(np.clip(scene + np.random.default_rng(1100 + c).normal(0.0, 2.0, scene.shape) + 0.5, 0, 255).astype(np.uint8)).to(dev) for c in range(2)]
kps = [torch.from_numpy(base[np.random.default_rng(2100 + c).permutation(N)].view(np.uint8).reshape(-1, 20).copy()).to(dev) for c in range(2)]
arena = torch.zeros((2, N, 64), dtype=torch.uint8, device=dev)
match = torch.zeros((N,), dtype=torch.int32, device=dev)
ip, kp, dp = [t.data_ptr() for t in imgs], [t.data_ptr() for t in kps], [arena[0].data_ptr(), arena[1].data_ptr()]

def make(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return Context(device=0, width=W, height=H, maxkp=N)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

base_ctx = make({})
def baseline():
    base_ctx.describe_batch_dev(ip, W, H, W, kp, [N, N], dp, s)
    base_ctx.match_2nn_dev(dp[0], N, dp[1], N, THR, match.data_ptr(), s)
def timed(fn, n=300):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for _ in range(400): baseline()
torch.cuda.synchronize()
want = match.clone()
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
configs = [("chunks=1", {}, 1), ("chunks=2", {}, 2), ("chunks=3", {}, 3), ("chunks=5", {}, 5)]
lists = ["2,1,1,1", "1,1,1,1,1", "2,2,1", "3,1,1", "2,1,2", "1,2,2", "3,2", "4,1", "1,1,3", "1,4"]
targets = ["0", "512", "384", "256"] if not quick else ["0", "384"]
for l in (lists if not quick else lists[:4]):
    for t in targets:
        configs.append(("list=%s target=%s" % (l, t), {"CLC_PAIR_CHUNKS": l, "CLC_PAIR_TARGET_BLOCKS": t}, 0))
print("baseline = clc_describe_batch_dev + clc_match_2nn_dev on one stream; us per step, three alternated legs each")
for name, env, chunks in configs:
    c = make(env)
    def pair():
        c.describe_match_pair_dev(ip, W, H, W, kp, [N, N], dp, THR, match.data_ptr(), chunks=chunks, stream=s)
    for _ in range(100): pair()
    torch.cuda.synchronize()
    same = bool(torch.equal(match, want))
    a, b = [], []
    for leg in range(3):
        b.append(timed(baseline)); a.append(timed(pair))
    same = same and bool(torch.equal(match, want))
    print("%-34s pair %6.2f %6.2f %6.2f   baseline %6.2f %6.2f %6.2f   identical %s" % (name, *a, *b, same), flush=True)
    c.close()
base_ctx.close()
