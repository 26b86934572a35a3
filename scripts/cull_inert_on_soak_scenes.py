"""The quadrant cull on the heavy-tailed soak's scenes (scripts/soak_heavy.py's generator, given seeds): the frame with the cull
on against the frame with every listed Gaussian evaluated against every live block (libmgs_debug.so's knob), bit for bit, under
both raster schedules.    python scripts/cull_inert_on_soak_scenes.py 48 67 ..."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import camera_ring, ops, synthetic_scene_heavy_tailed, _lib
DEV = "cuda"
def _t(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
for seed in [int(x) for x in sys.argv[1:]] or [48, 67]:
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(60_000, 400_000)); W = int(rng.integers(300, 1300)); H = int(rng.integers(200, 800)); deg = int(rng.integers(0, 4))
    g = synthetic_scene_heavy_tailed(n, math.log(float(rng.uniform(0.004, 0.03))), deg, seed, n_clusters=int(rng.integers(3, 120)),
                                     n_screen_filling=int(rng.integers(0, 9)), n_needles=int(rng.integers(0, n // 20)))
    cam = camera_ring(1, W, H, thetas=[float(rng.uniform(0, 6.28))], radius=float(rng.uniform(4, 9)))[0]
    t = g.to_torch(DEV, deg); vm, K = _t(cam.viewmat()), _t(cam.K)
    tw, th = -(-W // 16), -(-H // 16)
    radii, m2d, dep, con, _, feats, splats = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H,
                                                                       0.3, 0.01, 1e10, 0.0, False, True, want_splats=True)
    tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 60_000_000, want_tiles_per_gauss=False)
    for latency in (False, True):
        def frame():
            return ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, latency=latency,
                                         splats=splats, expected_last=True, channels=4)
        with _lib.use_debug_lib() as dbg:
            try:
                dbg.mgs_debug_set_raster_cull(1); a = frame()
                dbg.mgs_debug_set_raster_cull(0); b = frame()
            finally:
                dbg.mgs_debug_set_raster_cull(1)
        c = frame()
        print(f"seed {seed} ({n} Gaussians, {W}x{H}, per-block schedule {latency}): cull on == cull off:",
              all(torch.equal(x, y) for x, y in zip(a, b)), " shipped == debug:", all(torch.equal(x, y) for x, y in zip(a, c)),
              " pairs", int(tl.n_isect))
