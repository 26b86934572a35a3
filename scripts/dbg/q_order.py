"""EXPERIMENT: the per-block forward kernel with its tiles launched by falling list length (order from torch.argsort)."""
import math, sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, _lib
n, mu, W, H, deg = 1_000_000, 0.012, 1920, 1080, 3
dev = "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
radii, m2d, dep, con, _, feats, splats = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True)
tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 8_000_000, want_tiles_per_gauss=False, conics=con, opacities=t["opacities"])
lens = (tl.tile_offsets[1:] - tl.tile_offsets[:-1])
order = torch.argsort(lens, descending=True).to(torch.int32).contiguous()
# groups of four consecutive tiles by falling group total (what the binning's scatter kernel could hand over for free)
g4 = torch.nn.functional.pad(lens, (0, (-len(lens)) % 4)).view(-1, 4).sum(1)
go = torch.argsort(g4, descending=True)
order4 = (go[:, None] * 4 + torch.arange(4, device=dev)[None]).reshape(-1)
order4 = order4[order4 < len(lens)].to(torch.int32).contiguous()
L = _lib.lib()
L.mgs_debug_set_tile_order.argtypes = [ctypes.c_void_p]; L.mgs_debug_set_tile_order.restype = None
ref = None
for track in (False, True):
    for name, o in (("index order", None), ("by tile length", order), ("by group-of-4 total", order4)):
        L.mgs_debug_set_tile_order(o.data_ptr() if o is not None else None)
        out = None
        def run():
            return ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, out=out, track_last=track, splats=splats, expected_last=True, latency=True)
        for _ in range(5): out = run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30)
        if ref is None: ref = (out[0].clone(), out[1].clone())
        print(f"track_last={track} {name}: {best*1e3:.1f} us identical={torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])}")
L.mgs_debug_set_tile_order(None)
