"""Soak test (not part of pytest): random small scenes (anisotropic, faint or dense, camera inside or far,
every render mode, anti-aliased or not, both raster schedules) against the fp64 NumPy oracle through the
parity gate of the tests (oracle.gs_oracle_np.check_frame): counts the pixels over 1e-4 that NO threshold of
the oracle's blend explains -- there must be none."""
import math, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gs_oracle_np as O
from robosimgs_amd import rasterization, synthetic_scene, camera_ring
DEV = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
RULE = sys.argv[2] if len(sys.argv) > 2 else "classic"      # radius rule: "classic" | "opacity_aware" | "both" (by seed)
def _t(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
f32 = lambda m: np.asarray(m, dtype=np.float32).astype(np.float64)
unexplained = over = 0; worst_nonflip = 0.0; worst_flag = 0.0; failures = []
for seed in range(N):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(50, 3000)); W = int(rng.integers(20, 140)); H = int(rng.integers(20, 100)); deg = int(rng.integers(0, 4))
    g = synthetic_scene(n, math.log(float(rng.uniform(0.03, 0.6))), deg, seed)
    g.log_scales[:, int(rng.integers(0, 3))] += float(rng.uniform(-2, 2.0))
    g.opacity_logits[:] += float(rng.uniform(-3, 3))
    cam = camera_ring(1, W, H, thetas=[float(rng.uniform(0, 6.28))], radius=float(rng.uniform(1.0, 10)))[0]
    mode = str(rng.choice(["RGB", "RGB+ED", "RGB+D"])); aa = bool(rng.integers(0, 2)); sched = str(rng.choice(["latency", "throughput"]))
    t = g.to_torch(DEV, deg)
    rm_ = "antialiased" if aa else "classic"
    rule = RULE if RULE != "both" else ("classic", "opacity_aware")[seed & 1]
    c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], _t(cam.viewmat())[None], _t(cam.K)[None],
                               W, H, sh_degree=deg, render_mode=mode, rasterize_mode=rm_, raster_schedule=sched, radius_rule=rule,
                               tile_bounds=("tight", "classic")[(seed >> 1) & 1])
    ref, ra, rm = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, f32(cam.viewmat()), f32(cam.K), W, H, sh_degree=deg,
                           render_mode=mode, rasterize_mode=rm_, margins=True, flip_eps=O.EPS_PATH, radius_rule=rule)
    try:
        st = O.check_frame(c[0].cpu().numpy(), a[0].cpu().numpy(), ref, ra, rm["margins"], O.EPS_PATH, rm["edge_mask"],
                           expected_depth="E" in mode, max_explained=1.0, what=f"seed {seed}",
                           flip_weight=rm["flip_weight"], feat_max=rm["feat_max"], require_flip_bound=True)
    except AssertionError as e:
        failures.append(str(e)[:300]); continue
    over += st["over_tol"]; worst_nonflip = max(worst_nonflip, st["max_err_over_tol_nonflip"]); worst_flag = max(worst_flag, st["could_flip_frac"])
print(f"{N} scenes (radius rule {RULE}): pixels over 1e-4: {over}, scenes with an unexplained pixel: {len(failures)}, worst error off the thresholds "
      f"{worst_nonflip:.3f} x tolerance, largest could-flip fraction {worst_flag:.4f}")
for f in failures: print("FAIL", f)
