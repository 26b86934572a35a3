"""Timing of the tile-raster backward (record path) + reduce at config 2, 4 channels (HIP events)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
n, mu, W, H, deg = 1_000_000, 0.012, 1920, 1080, 3
dev = "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0)
if os.environ.get("MORTON", "1") != "0":      # the order FrameRenderer keeps its resident scene in
    g = g.sorted_by_locality()
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
radii, m2d, dep, con, _, feats, splats = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True)
tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 4_700_000, want_tiles_per_gauss=False, want_pair_info=True, conics=con, opacities=t["opacities"])
out = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats, latency=True)
torch.manual_seed(0)
vr = torch.rand(H, W, 4, device=dev); va = torch.rand(H, W, device=dev)
def run():
    return ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl, out[1], out[2], vr, va, splats=splats)
for _ in range(3): r = run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(15): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 15)
print(f"{os.environ.get('TAG', '')} raster_bwd_det (memset + raster_bwd + reduce): {best*1e3:.1f} us   checksum {float(r[0].double().abs().sum()):.6e} {float(r[2].double().abs().sum()):.6e}")
