import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from oracle import cpu_ref, gs_oracle_np as O
import test_gpu_heavy as T
from robosimgs_amd import ops
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 67
g, cam, W, H, deg = T._soak_scene(seed)
t, radii, m2d, dep, con, feats, splats, tl, tw, th = T._stage(g, cam, W, H, deg)
r, a, _ = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats, group_order=tl.group_order)
got, ga = r.cpu().numpy(), a.cpu().numpy()
vm32, K32 = np.asarray(cam.viewmat(), np.float32), np.asarray(cam.K, np.float32)
ref, ra, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vm32, K32, W, H, deg, with_depth=True, flip_eps=O.EPS_PATH, want_projected=True)
r32, a32, _i = cpu_ref.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vm32, K32, W, H, deg, with_depth=True)
for name, x, xa in (("HIP", got, ga), ("fp32 port", r32, a32)):
    d = np.abs(x - ref)
    print(name, "pixels over 1e-4 per channel:", [(int((d[..., c] > 1e-4).sum())) for c in range(4)], "alpha:", int((np.abs(xa - ra) > 1e-4).sum()))
# projection accuracy: GPU vs fp64 port, numpy fp32 vs fp64
vis = info["radii"] > 0
M2, CO = m2d.cpu().numpy().astype(np.float64), con.cpu().numpy().astype(np.float64)
p32 = O.project(g.means, g.quats, g.scales, vm32.astype(np.float64), K32.astype(np.float64), W, H, dtype=np.float32)
for name, mm, cc in (("HIP", M2, CO), ("numpy fp32", p32["means2d"].astype(np.float64), p32["conics"].astype(np.float64))):
    em = np.abs(mm - info["means2d"])[vis].max(axis=1)
    ec = (np.abs(cc - info["conics"])[vis] / (np.abs(info["conics"][vis]).max(axis=1, keepdims=True) + 1e-30)).max(axis=1)
    print(f"{name}: means2d error px: median {np.median(em):.2e} p99 {np.quantile(em, .99):.2e} max {em.max():.2e}; conic rel error: median {np.median(ec):.2e} p99 {np.quantile(ec, .99):.2e} p99.9 {np.quantile(ec, .999):.2e} max {ec.max():.2e}")
c = info["conics"][vis]
tr, det = c[:, 0] + c[:, 2], c[:, 0] * c[:, 2] - c[:, 1] ** 2
disc = np.sqrt(np.maximum(tr * tr / 4 - det, 0))
kappa = (tr / 2 + disc) / np.maximum(tr / 2 - disc, 1e-300)
for name, cc in (("HIP", CO), ("numpy fp32", p32["conics"].astype(np.float64))):
    ec = (np.abs(cc - info["conics"])[vis] / (np.abs(info["conics"][vis]).max(axis=1, keepdims=True) + 1e-30)).max(axis=1)
    print(name, "conic rel error / kappa: p99 %.2e max %.2e; kappa max %.2e; rel error / (2e-5 + 1e-7 kappa) max %.3f" % (np.quantile(ec / kappa, .99), (ec / kappa).max(), kappa.max(), (ec / (2e-5 + 1e-7 * kappa)).max()))
