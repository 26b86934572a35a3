"""Soak test (not part of pytest): 40 random small scenes (anisotropic, faint or dense, camera inside or far,
every render mode, anti-aliased or not) against the fp64 NumPy oracle; reports the worst fraction of pixels off by
more than 1e-4."""
import math, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import gs_oracle_np as O
from robosimgs_amd import rasterization, synthetic_scene, camera_ring
DEV="cuda"
def _t(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
worst=0; bad_scenes=0
for seed in range(40):
    rng=np.random.default_rng(9000+seed)
    n=int(rng.integers(50,3000)); W=int(rng.integers(20,140)); H=int(rng.integers(20,100)); deg=int(rng.integers(0,4))
    g=synthetic_scene(n, math.log(float(rng.uniform(0.03,0.6))), deg, seed)
    g.log_scales[:, int(rng.integers(0,3))]+=float(rng.uniform(-2,2.0))
    g.opacity_logits[:]+=float(rng.uniform(-3,3))
    cam=camera_ring(1,W,H,thetas=[float(rng.uniform(0,6.28))], radius=float(rng.uniform(1.0,10)))[0]
    mode=str(rng.choice(["RGB","RGB+ED","RGB+D"])); aa=bool(rng.integers(0,2))
    t=g.to_torch(DEV,deg)
    c,a,meta=rasterization(t["means"],t["quats"],t["scales"],t["opacities"],t["colors"],_t(cam.viewmat())[None],_t(cam.K)[None],W,H,sh_degree=deg,render_mode=mode,rasterize_mode="antialiased" if aa else "classic")
    ref,ra,rm=O.render(g.means,g.quats,g.scales,g.opacities,g.sh_coeffs,cam.viewmat(),cam.K,W,H,sh_degree=deg,render_mode=mode,rasterize_mode="antialiased" if aa else "classic")
    d=np.abs(c[0].cpu().numpy()-ref); 
    if "E" in mode: d[..., -1] = d[..., -1] / np.maximum(np.abs(ref[..., -1]), 1.0) * 0.05   # expected depth: relative
    frac=float((d.max(-1)>1e-4).mean()); da=float(np.abs(a[0,...,0].cpu().numpy()-ra[...,0]).max())
    worst=max(worst,frac)
    if frac>2e-3 or not np.isfinite(c.cpu().numpy()).all(): bad_scenes+=1; print("SCENE",seed,n,W,H,deg,mode,aa,"frac",frac,"alpha max",da)
print("forward-vs-oracle sweeps done; scenes over 0.2 % bad pixels:",bad_scenes,"worst fraction",worst)
