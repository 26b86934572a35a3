"""What would 16-lane (4x4-pixel) culled queues buy the forward raster?  (round-2 review, item 2c)
For a sample of the bench scene's visible Gaussians: the pixels each one reaches with alpha >= 1/255 inside the image,
the 8x8 blocks and the 4x4 blocks those pixels fall into.  A block evaluation costs its lane count whatever the
number of lanes that pass, so   lane utilisation = pixels / (blocks x lanes per block).
A wave of four 4x4-block queues runs as long as its longest queue: `lockstep` prices that on the same sample by
grouping the 4x4 blocks of an 8x8 block (the four sub-queues of a 16-lane-group wave cover 2 x 2 of them each)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
n, W, H, deg = 1_000_000, 1920, 1080, 3
dev = "cuda"
g = synthetic_scene(n, math.log(0.012), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev); K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
radii, m2d, dep, con, _, feats, splats = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True)
vis = torch.nonzero(radii > 0).flatten()
sel = vis[torch.randperm(len(vis), device=dev, generator=torch.Generator(dev).manual_seed(0))[:60000]]
mx, my = m2d[sel, 0], m2d[sel, 1]
a, b, c, op, r = con[sel, 0], con[sel, 1], con[sel, 2], t["opacities"][sel], radii[sel].float()
order = torch.argsort(r)
tot_px = tot_b8 = tot_b4 = tot_lock = 0
for lo in range(0, len(sel), 1000):
    idx = order[lo:lo + 1000]
    R = int(min(r[idx].max().item(), 200))
    off = torch.arange(-R, R + 1, device=dev, dtype=torch.float32)
    px = torch.floor(mx[idx])[:, None] + off[None, :]          # pixel columns  [n, G]
    py = torch.floor(my[idx])[:, None] + off[None, :]
    dx = (px + 0.5) - mx[idx][:, None]; dy = (py + 0.5) - my[idx][:, None]
    sig = 0.5 * (a[idx][:, None, None] * dx[:, None, :] ** 2 + c[idx][:, None, None] * dy[:, :, None] ** 2) + b[idx][:, None, None] * dx[:, None, :] * dy[:, :, None]
    alpha = torch.clamp(op[idx][:, None, None] * torch.exp(-sig), max=0.999)
    inside = (px[:, None, :] >= 0) & (px[:, None, :] < W) & (py[:, :, None] >= 0) & (py[:, :, None] < H)
    inside = inside & (dx[:, None, :].abs() <= r[idx][:, None, None]) & (dy[:, :, None].abs() <= r[idx][:, None, None])   # the 3-sigma square of A.2 step 7
    hit = (alpha >= 1.0 / 255.0) & (sig >= 0) & inside                               # [n, Gy, Gx]
    X = px[:, None, :].expand_as(hit).long(); Y = py[:, :, None].expand_as(hit).long()
    gi = torch.arange(len(idx), device=dev)[:, None, None].expand_as(hit)
    k8 = (gi[hit] << 40) | ((Y[hit] >> 3) << 20) | (X[hit] >> 3)
    k4 = (gi[hit] << 40) | ((Y[hit] >> 2) << 20) | (X[hit] >> 2)
    u8 = torch.unique(k8); u4 = torch.unique(k4)
    tot_px += int(hit.sum()); tot_b8 += len(u8); tot_b4 += len(u4)
    # lockstep: per (Gaussian, 8x8 block) the wave of four 2x2-sub-block queues... price a 16x16 tile handled by FOUR waves of
    # four 16-lane groups: each 8x8 block is one wave, its four 4x4 blocks are the groups; the wave's trips = entries of its
    # longest group queue, so a Gaussian that reaches any 4x4 block of the 8x8 block costs the wave at least its share;
    # lower bound on lane-evaluations: 64 x (number of 8x8 blocks reached) x (4x4 blocks reached in it) / 4, summed --
    # equal to 16 x 4x4-blocks only if the four queues were equally long.
print(f"sampled Gaussians {len(sel)}: pixels reached {tot_px / len(sel):.1f} each; 8x8 blocks {tot_b8 / len(sel):.2f}; 4x4 blocks {tot_b4 / len(sel):.2f}")
print(f"lane-evaluations per Gaussian: 8x8 queues {64 * tot_b8 / len(sel):.0f} (utilisation {tot_px / (64 * tot_b8):.3f}); 4x4 queues {16 * tot_b4 / len(sel):.0f} (utilisation {tot_px / (16 * tot_b4):.3f}) if the four queues of a wave were equally long")
