"""Per-tile forward raster alone against KiB of unused dynamic LDS per workgroup (debug library, raster_opts bits 8..):
where the time steps up tells how many workgroups of 16.6 KB + pad a CU really holds."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MGS_USE_DEBUG_LIB"] = "1"
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, _lib
n, mu, W, H, deg, dev = 1_000_000, 0.012, 1920, 1080, 3, "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0).sorted_by_locality()
t = g.to_torch(dev, deg)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev); K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
radii, m2d, dep, con, _, feats, splats, seed = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight")
tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 4_700_000, want_tiles_per_gauss=False, seed=seed, want_tile_ids=False)
L = _lib.lib()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for lat in (False, True):
    for pad in [int(x) for x in os.environ.get("PADS", "0,2,4,6,8,10,12,16,20,24,36").split(",")]:
        L.mgs_debug_set_raster_opts(3 | (pad << 8))
        f = lambda: ops.rasterize_fwd_raw(None, None, None, None, None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, track_last=False, splats=splats,
                                          expected_last=True, latency=lat, group_order=tl.group_order, channels=4)
        for _ in range(5): f()
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        print(f"{'per block' if lat else 'per tile '}  pad {pad:2d} KiB: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us")
L.mgs_debug_set_raster_opts(3)
