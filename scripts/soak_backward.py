"""Soak test (not part of pytest): random small scenes, whole-path gradients against autograd of the fp64 torch
oracle through the gate of tests/grad_gate.py: cosine, fraction of rows over 5e-3 and -- where the fp64 port can price
them (not anti-aliased) -- EVERY row within rounding + 1.5 x its flip budget.  The backward walks 64-entry segments, 256-entry
segments or the whole list, in turn."""
import math, sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import gs_oracle_torch as OT, gs_oracle_np as O
from robosimgs_amd import rasterization, synthetic_scene, camera_ring
from grad_gate import compare, oracle_budgets, parameter_budgets
DEV = "cuda"
def _t(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
def _d(a, g=False): return torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=g)
worst_cos, worst_frac, worst_ratio, failures, gated = 1.0, 0.0, 0.0, 0, 0
N = int(os.environ.get("SCENES", 24))
for seed in range(N):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(300, 6000)); W = int(rng.integers(32, 160)); H = int(rng.integers(32, 120)); deg = int(rng.integers(0, 4))
    g = synthetic_scene(n, math.log(float(rng.uniform(0.05, 0.4))), deg, seed)
    g.log_scales[:, int(rng.integers(0, 3))] += float(rng.uniform(-1.5, 1.5))
    g.opacity_logits[:] += float(rng.uniform(-3, 4))          # from faint (long walks) up to very opaque (the 0.999 clamp)
    cam = camera_ring(1, W, H, thetas=[float(rng.uniform(0, 6.28))], radius=float(rng.uniform(3, 9)))[0]
    mode = str(rng.choice(["RGB", "RGB+ED", "RGB+D"])); aa = bool(rng.integers(0, 3) == 0)
    seg = [64, 256, 0][seed % 3]
    rm = "antialiased" if aa else "classic"
    t = g.to_torch(DEV, deg)
    names = ["means", "quats", "scales", "opacities", "colors"]
    for k in names: t[k].requires_grad_(True)
    c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], _t(cam.viewmat())[None],
                               _t(cam.K)[None], W, H, sh_degree=deg, render_mode=mode, rasterize_mode=rm, backward_segment=seg)
    wr, wa = rng.normal(size=tuple(c.shape[1:])), rng.normal(size=(H, W))
    ((c[0] * _t(wr)).sum() + (a[0, ..., 0] * _t(wa)).sum()).backward()
    r = {"means": _d(g.means, True), "quats": _d(g.quats, True), "scales": _d(g.scales, True),
         "opacities": _d(g.opacities, True), "colors": _d(g.sh_coeffs[:, :(deg + 1) ** 2], True)}
    img, al, _ = OT.render(r["means"], r["quats"], r["scales"], r["opacities"], r["colors"], _d(cam.viewmat()), _d(cam.K),
                           W, H, sh_degree=deg, render_mode=mode, rasterize_mode=rm)
    ((img * _d(wr)).sum() + (al[..., 0] * _d(wa)).sum()).backward()
    budgets = {k: None for k in names}
    if not aa:
        f32 = lambda m: np.asarray(m, dtype=np.float32)
        info = oracle_budgets(g, f32(cam.viewmat()), f32(cam.K), W, H, deg, mode, wr, wa, O.EPS_PATH_GRAD)
        budgets.update(parameter_budgets(g, f32(cam.viewmat()), f32(cam.K), W, H, deg, mode != "RGB", info["budget"]))
        budgets["opacities"] = info["budget"][:, 3]
        gated += 1
    for k in names:
        try:
            st = compare(f"scene {seed} {k}", t[k].grad, r[k].grad.numpy().reshape(n, -1), row_tol=5e-3, bad_frac=2e-2, cos_min=0.999,
                         budget=budgets[k], verbose=False)
            worst_cos, worst_frac = min(worst_cos, st["cosine"]), max(worst_frac, st["rows_over_tol"] / st["rows"])
            worst_ratio = max(worst_ratio, st.get("worst_ratio", 0.0))
        except AssertionError as e:
            failures += 1
            print("SCENE", seed, n, W, H, deg, mode, "aa" if aa else "", "segment", seg, k, str(e)[:200])
print(f"backward-vs-autograd sweeps done: {N} scenes ({gated} with flip budgets), failures {failures}; worst cosine {worst_cos:.8f}, worst "
      f"fraction of rows over 5e-3 {worst_frac:.4f}, worst |d| / (rounding + 1.5 budget) {worst_ratio:.3f}")
