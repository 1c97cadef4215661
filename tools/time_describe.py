import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, synth
from coloc_amd import Context
W,H,N=640,480,10000
ctx=Context(device=0,width=W,height=H,maxkp=N)
dev=torch.device("cuda",0)
def run(name, img, kps):
    ctx.pyramid_build(img)
    dk=torch.from_numpy(kps.view(np.uint8).reshape(-1,20).copy()).to(dev)
    dd=torch.empty((N,64),dtype=torch.uint8,device=dev)
    st=torch.cuda.Stream(); torch.cuda.set_stream(st); s=st.cuda_stream
    for _ in range(3): ctx.describe_dev(dk.data_ptr(),N,dd.data_ptr(),s)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(20):
        e0.record(); ctx.describe_dev(dk.data_ptr(),N,dd.data_ptr(),s); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1)*1e3)
    print("%-40s median %.1f us min %.1f us"%(name, sorted(ts)[10], min(ts)))
rect=synth.rect_image(W,H,seed=1000,noise_sigma=2.0)
rnd=np.random.default_rng(0).integers(0,256,(H,W),dtype=np.uint8)
k_area=synth.random_keypoints(N,W,H,seed=2000)
k_uni=k_area.copy(); rng=np.random.default_rng(1); k_uni["scale"]=rng.integers(0,8,N)
ws,hs,_=synth.pyramid_dims(W,H)
k_uni["x"]=(3+rng.random(N)*(np.array(ws)[k_uni["scale"]]-7)).astype(np.int32); k_uni["y"]=(3+rng.random(N)*(np.array(hs)[k_uni["scale"]]-7)).astype(np.int32)
k_l0=k_area.copy(); k_l0["scale"]=0; k_l0["x"]=(3+rng.random(N)*(W-7)).astype(np.int32); k_l0["y"]=(3+rng.random(N)*(H-7)).astype(np.int32)
k_l7=k_area.copy(); k_l7["scale"]=7; k_l7["x"]=(3+rng.random(N)*(ws[7]-7)).astype(np.int32); k_l7["y"]=(3+rng.random(N)*(hs[7]-7)).astype(np.int32)
k_a0=k_area.copy(); k_a0["angle"]=0
k_sorted=k_area[np.lexsort((k_area["x"],k_area["y"],k_area["scale"]))]
def run_rebuild(name,img,kps):
    dimg=torch.from_numpy(img).to(dev)
    dk=torch.from_numpy(kps.view(np.uint8).reshape(-1,20).copy()).to(dev)
    dd=torch.empty((N,64),dtype=torch.uint8,device=dev)
    st=torch.cuda.Stream(); torch.cuda.set_stream(st); s=st.cuda_stream
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(20):
        ctx.pyramid_build_dev(dimg.data_ptr(),W,H,W,s); ctx.describe_dev(dk.data_ptr(),N,dd.data_ptr(),s)
    torch.cuda.synchronize(); ctx.profile_enable(False)
    p=ctx.profile_read(); print(name, {k:(round(v[0]/max(v[1],1)*1e3,1),v[1]) for k,v in p.items() if v[1]})
run_rebuild("rebuild pyramid each time", rect, k_area)
for nm,img,k in [("rect area-weighted",rect,k_area),("random-img area-weighted",rnd,k_area),("rect uniform-scale",rect,k_uni),("rect level0 only",rect,k_l0),("rect level7 only",rect,k_l7),("rect angle=0",rect,k_a0),("rect sorted (scale,y,x)",rect,k_sorted)]:
    run(nm,img,k)
