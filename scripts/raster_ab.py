"""A/B timing of the tile-raster forward kernel (config 2, 4 channels, inference variant) under the
scheduling knobs of mgs_debug_set_raster_opts.  usage: raster_ab.py [opts ...]"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, _lib
opts_list = [int(x, 0) for x in sys.argv[1:]] or [1, 5]     # 1: one wave per tile (+ priority), 5: one wave per 8x8 block
n, mu, W, H, deg = int(os.environ.get("N", 1_000_000)), float(os.environ.get("MU", 0.012)), int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), 3
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
tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 40_000_000 if n > 2_000_000 else 8_000_000, want_tiles_per_gauss=False, conics=con, opacities=t["opacities"])
lens = (tl.tile_offsets[1:] - tl.tile_offsets[:-1]).float()
print("n_isect", int(tl.n_isect), "tile list length: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (lens.mean(), lens.median(), lens.quantile(0.9), lens.quantile(0.99), lens.max()))
ref = None
os.environ["MGS_USE_DEBUG_LIB"] = "1"      # the scheduling knobs exist in libmgs_debug.so only
L = _lib.lib()
for track in (False, True):
    for o in opts_list:
        L.mgs_debug_set_raster_opts(o)
        out = None
        def run():
            return ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, out=out, track_last=track, splats=splats, expected_last=True, group_order=tl.group_order)
        for _ in range(5): out = run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30)
        if ref is None: ref = (out[0].clone(), out[1].clone())
        same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
        print(f"track_last={track} opts={o}: {best*1e3:.1f} us  identical={same}")
L.mgs_debug_set_raster_opts(3)
