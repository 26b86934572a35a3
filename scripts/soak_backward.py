"""Soak test (not part of pytest): random small scenes, whole-path gradients against autograd of the fp64 torch
oracle (scaled row error and cosine per parameter group)."""
import math, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import gs_oracle_torch as OT
from robosimgs_amd import rasterization, synthetic_scene, camera_ring
DEV = "cuda"
def _t(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
def _d(a, g=False): return torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=g)
worst_cos, worst_frac = 1.0, 0.0
for seed in range(12):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(300, 2500)); W = int(rng.integers(32, 100)); H = int(rng.integers(32, 80)); deg = int(rng.integers(0, 4))
    g = synthetic_scene(n, math.log(float(rng.uniform(0.05, 0.4))), deg, seed)
    g.log_scales[:, int(rng.integers(0, 3))] += float(rng.uniform(-1.5, 1.5))
    g.opacity_logits[:] += float(rng.uniform(-2, 4))          # up to very opaque: exercises the 0.999 clamp
    cam = camera_ring(1, W, H, thetas=[float(rng.uniform(0, 6.28))], radius=float(rng.uniform(3, 9)))[0]
    mode = str(rng.choice(["RGB", "RGB+ED", "RGB+D"])); aa = bool(rng.integers(0, 2))
    rm = "antialiased" if aa else "classic"
    t = g.to_torch(DEV, deg)
    names = ["means", "quats", "scales", "opacities", "colors"]
    for k in names: t[k].requires_grad_(True)
    c, a, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], _t(cam.viewmat())[None],
                            _t(cam.K)[None], W, H, sh_degree=deg, render_mode=mode, rasterize_mode=rm)
    wr, wa = rng.normal(size=tuple(c.shape[1:])), rng.normal(size=(H, W))
    ((c[0] * _t(wr)).sum() + (a[0, ..., 0] * _t(wa)).sum()).backward()
    r = {"means": _d(g.means, True), "quats": _d(g.quats, True), "scales": _d(g.scales, True),
         "opacities": _d(g.opacities, True), "colors": _d(g.sh_coeffs[:, :(deg + 1) ** 2], True)}
    img, al, _ = OT.render(r["means"], r["quats"], r["scales"], r["opacities"], r["colors"], _d(cam.viewmat()), _d(cam.K),
                           W, H, sh_degree=deg, render_mode=mode, rasterize_mode=rm)
    ((img * _d(wr)).sum() + (al[..., 0] * _d(wa)).sum()).backward()
    for k in names:
        got = t[k].grad.detach().cpu().double().numpy().reshape(n, -1); ref = r[k].grad.numpy().reshape(n, -1)
        scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-3 * np.abs(ref).max() + 1e-30
        frac = float(((np.abs(got - ref) / scale).max(axis=1) > 5e-3).mean())
        cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
        worst_cos, worst_frac = min(worst_cos, cos), max(worst_frac, frac)
        if cos < 0.999 or frac > 2e-2 or not np.isfinite(got).all():
            print("SCENE", seed, n, W, H, deg, mode, aa, k, "cos", cos, "rows off", frac)
print("backward-vs-autograd sweeps done: worst cosine", worst_cos, "worst fraction of rows over 5e-3", worst_frac)
