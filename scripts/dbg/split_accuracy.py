"""Which walk is closer to the fp64 blend backward (oracle/gs_cpu.cpp) at configs[2]: whole list or segments?"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
from oracle import cpu_ref
n, mu, W, H, deg = 1_000_000, 0.012, 1920, 1080, 3
dev = "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vmf, Kf = cam.viewmat().astype(np.float32), cam.K.astype(np.float32)
vm, K = torch.from_numpy(vmf).to(dev), torch.from_numpy(Kf).to(dev)
tw, th = -(-W // 16), -(-H // 16)
CAP = 4_700_000
radii, m2d, dep, con, _, feats, splats = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True)
tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, CAP, want_tiles_per_gauss=False, want_pair_info=True, conics=con, opacities=t["opacities"])
rng = np.random.default_rng(21)
w_img = rng.normal(size=(H, W, 4)).astype(np.float32); w_a = rng.normal(size=(H, W)).astype(np.float32)
vr, va = torch.from_numpy(w_img).to(dev), torch.from_numpy(w_a).to(dev)
_, _, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vmf, Kf, W, H, deg, with_depth=True,
                                margins=False, v_render=w_img, v_alpha=w_a)
ref = [info["g_means2d"], info["g_conics"], info["g_feats"], info["g_opacities"].reshape(-1, 1)]
for seg in (0, 64, 128, 256):
    ck = ops.checkpoint_buffer(CAP, tw, th, 4, seg, dev) if seg else None
    out = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats,
                                latency=True, group_order=tl.group_order, channels=4, checkpoints=ck, checkpoint_interval=seg)
    r = ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl, out[1], out[2], vr, va, splats=splats,
                                  render_out=out[0] if seg else None, checkpoints=ck, checkpoint_interval=seg)
    line = f"segment {seg:4d}:"
    for name, got, rf in zip(("means2d", "conics", "feats", "opacities"), r[:4], ref):
        got = got.double().cpu().numpy().reshape(rf.shape)
        scale = np.abs(rf).max(axis=1, keepdims=True) + 1e-3 * np.abs(rf).max()
        err = (np.abs(got - rf) / scale).max(axis=1)
        line += f"  {name}: rows>2e-3 {int((err > 2e-3).sum())}, p99.9 {np.quantile(err, 0.999):.2e}, rms {np.sqrt((err ** 2).mean()):.2e}, max|d|/max|ref| {np.abs(got - rf).max() / np.abs(rf).max():.2e}"
    print(line, flush=True)
