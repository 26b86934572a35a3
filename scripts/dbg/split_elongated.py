"""Needle-like footprints: whole-list walk vs segmented walk against fp64 autograd (oracle/gs_oracle_torch.py)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, rasterization
from oracle import gs_oracle_torch as OT
DEV = "cuda"
W, H = 200, 136
rng = np.random.default_rng(7)
for case in ("elongated", "huge", "config1"):
    g = synthetic_scene(6000, math.log(0.05), 1, 11)
    if case == "elongated":
        g.log_scales[:, 0] += 2.5; g.log_scales[:, 1:] -= 2.0
    elif case == "huge":
        g.log_scales[::50] += 3.5
    cam = camera_ring(1, W, H, thetas=[0.7])[0]
    t = g.to_torch(DEV, 1)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    vm, K = f(cam.viewmat())[None], f(cam.K)[None]
    w_img = rng.normal(size=(H, W, 4)).astype(np.float32); w_a = rng.normal(size=(H, W, 1)).astype(np.float32)
    names = ("means", "quats", "scales", "opacities", "colors")
    res = {}
    for seg in (0, 64, 128):
        p = {k: t[k].clone().requires_grad_(True) for k in names}
        c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, W, H, sh_degree=1,
                                   render_mode="RGB+D", tile_bounds="classic", backward_segment=seg)
        ((c[0] * f(w_img)).sum() + (a[0] * f(w_a)).sum()).backward()
        res[seg] = {k: p[k].grad.double().cpu().numpy() for k in names}
    d = lambda x, grad=False: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=grad)
    r = {"means": d(g.means, True), "quats": d(g.quats, True), "scales": d(g.scales, True), "opacities": d(g.opacities, True),
         "colors": d(g.sh_coeffs[:, :4], True)}
    img, al, _ = OT.render(r["means"], r["quats"], r["scales"], r["opacities"], r["colors"], d(cam.viewmat().astype(np.float32)),
                           d(cam.K.astype(np.float32)), W, H, sh_degree=1, render_mode="RGB+D")
    ((img * d(w_img)).sum() + (al * d(w_a)).sum()).backward()
    print(f"== {case}: list max {int((meta['tile_lists'][0].tile_offsets[1:] - meta['tile_lists'][0].tile_offsets[:-1]).max())}")
    for k in names:
        ref = r[k].grad.numpy().reshape(res[0][k].shape)
        sc = np.abs(ref).max()
        line = f"  {k:10s} max|ref| {sc:9.3e}:"
        for seg in (0, 64, 128):
            e = np.abs(res[seg][k] - ref)
            line += f"  seg {seg}: max {e.max() / sc:.2e} rms {np.sqrt((e ** 2).mean()) / sc:.2e}"
        line += f"   whole vs seg64 max {np.abs(res[0][k] - res[64][k]).max() / sc:.2e}"
        print(line, flush=True)
