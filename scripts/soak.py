"""Soak test (not part of pytest): 300 random binning inputs bit-exact against the stable-sort formulation and 40
random scenes whose tightened tile lists must render and differentiate bit-identically to the classic ones."""
import math, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import gs_oracle_np as O
from robosimgs_amd import ops, rasterization, synthetic_scene, camera_ring
DEV="cuda"
def _t(a, dt=torch.float32): return torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(DEV)
bad=0
for seed in range(300):
    rng=np.random.default_rng(1000+seed)
    n=int(rng.integers(1, 40000)); tw=int(rng.integers(1, 300)); th=int(rng.integers(1, 40))
    w,h=tw*16-int(rng.integers(0,16)), th*16-int(rng.integers(0,16))
    m=np.column_stack([rng.uniform(-40,w+40,n), rng.uniform(-40,h+40,n)]).astype(np.float32)
    r=rng.choice([0,0,1,2,5,9,17,33,120], size=n).astype(np.int32)
    if n*30>2_000_000: r=np.minimum(r,17)
    d=rng.uniform(0.5,30,n).astype(np.float32); d[rng.integers(0,n,n//3)]=np.float32(3.5)
    tpg,ids,flat=ops.isect_tiles(_t(m)[None], torch.from_numpy(r).to(DEV)[None], _t(d)[None], 16, tw, th)
    rt,ri,rf=O.isect_tiles(m,r,d,16,tw,th,dtype=np.float32)
    ok=np.array_equal(tpg[0].cpu().numpy(),rt) and np.array_equal(ids.cpu().numpy(),ri) and np.array_equal(flat.cpu().numpy(),rf)
    if not ok: bad+=1; print("BINNING MISMATCH seed",seed,n,tw,th)
print("binning sweeps done, mismatches:",bad)
bad=0
for seed in range(40):
    rng=np.random.default_rng(5000+seed)
    n=int(rng.integers(200,20000)); W=int(rng.integers(17,400)); H=int(rng.integers(17,300)); deg=int(rng.integers(0,4))
    g=synthetic_scene(n, math.log(float(rng.uniform(0.01,0.4))), deg, seed)
    g.log_scales[:, int(rng.integers(0,3))]+=float(rng.uniform(-2,2.5))
    g.opacity_logits[:]+=float(rng.uniform(-4,2))
    cam=camera_ring(1,W,H,thetas=[float(rng.uniform(0,6.28))], radius=float(rng.uniform(2,12)))[0]
    t=g.to_torch(DEV,deg); vm,K=_t(cam.viewmat())[None],_t(cam.K)[None]
    outs=[]
    for b in ("classic","tight"):
        p={k:t[k].clone().requires_grad_(True) for k in ("means","quats","scales","opacities","colors")}
        # (backward_segment=0: bit identity of the gradients is a property of the whole-list walk; the segmented walk is soaked in soak_backward.py)
        c,a,meta=rasterization(p["means"],p["quats"],p["scales"],p["opacities"],p["colors"],vm,K,W,H,sh_degree=deg,render_mode="RGB+ED",tile_bounds=b,backward_segment=0)
        (c.sum()+a.sum()).backward()
        outs.append((c.detach(),a.detach(),[v.grad for v in p.values()],int(meta["n_isects"][0])))
    same=torch.equal(outs[0][0],outs[1][0]) and torch.equal(outs[0][1],outs[1][1]) and all(torch.equal(x,y) for x,y in zip(outs[0][2],outs[1][2]))
    fin=all(torch.isfinite(x).all() for x in outs[1][2]) and torch.isfinite(outs[1][0]).all()
    if not (same and fin): bad+=1; print("TIGHT MISMATCH seed",seed,n,W,H,deg,outs[0][3],outs[1][3], same, fin)
print("tight-vs-classic sweeps done, mismatches:",bad)
