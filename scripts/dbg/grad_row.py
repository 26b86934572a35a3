"""Diagnose one gradient row that the touched-gate rejects (tests/test_gpu_backward.py end-to-end, deg 3 RGB)."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gs_oracle_torch as OT, cpu_ref, gs_oracle_np as O
from robosimgs_amd import camera_ring, synthetic_scene, rasterization
DEV = "cuda"
_t = lambda a, g=False: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).requires_grad_(g)
_d = lambda a, g=False: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=g)
deg, mode, w, h = 3, "RGB", 112, 80
g = synthetic_scene(6000, math.log(0.07), deg, 0); cam = camera_ring(1, w, h, thetas=[0.3])[0]
t = g.to_torch(DEV, deg)
names = ["means", "quats", "scales", "opacities", "colors"]
for k in names: t[k].requires_grad_(True)
colors, alphas, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], _t(cam.viewmat()[None]), _t(cam.K[None]), w, h, sh_degree=deg, render_mode=mode)
rng = np.random.default_rng(2)
wr, wa = rng.normal(size=tuple(colors.shape[1:])), rng.normal(size=(h, w))
((colors[0] * _t(wr)).sum() + (alphas[0, ..., 0] * _t(wa)).sum()).backward()
r = {k: _d(v, True) for k, v in (("means", g.means), ("quats", g.quats), ("scales", g.scales), ("opacities", g.opacities), ("colors", g.sh_coeffs[:, :(deg + 1) ** 2]))}
img, al, rmeta = OT.render(r["means"], r["quats"], r["scales"], r["opacities"], r["colors"], _d(cam.viewmat()), _d(cam.K), w, h, sh_degree=deg, render_mode=mode)
((img * _d(wr)).sum() + (al[..., 0] * _d(wa)).sum()).backward()
f32 = lambda m: np.asarray(m, dtype=np.float32)
_, _, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, f32(cam.viewmat()), f32(cam.K), w, h, deg, flip_eps=O.EPS_PATH_GRAD, want_touched=True, want_projected=True, v_render=wr.astype(np.float32), v_alpha=wa.astype(np.float32))
got = t["opacities"].grad.cpu().double().numpy(); ref = r["opacities"].grad.numpy()
scale = np.abs(ref) + 1e-3 * np.abs(ref).max()
err = np.abs(got - ref) / scale
for row in np.argsort(-err)[:6]:
    print("row", row, "err", err[row], "got", got[row], "ref", ref[row], "port", info["g_opacities"][row], "max|ref|", np.abs(ref).max(), "opacity", g.opacities[row],
          "touched", info["touched"][row], "radius", info["radii"][row], "mean2d", info["means2d"][row], "conic", info["conics"][row])
