"""Run one stage of the config-2 frame repeatedly (for rocprofv3 --pmc passes).
usage: run_stage.py {raster|raster_bwd|binning|project|frame} [reps]"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, _lib
if os.environ.get("VARIANT_LIB"):      # another tree's libmgs.so (scripts/dbg/tsort_main_counters.sh)
    _lib._lib = _lib._load(os.path.abspath(os.environ["VARIANT_LIB"]))
stage = sys.argv[1] if len(sys.argv) > 1 else "raster"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, mu, W, H, deg = int(os.environ.get("N", 1_000_000)), float(os.environ.get("MU", 0.012)), int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080)), 3
CH = int(os.environ.get("CH", 4))      # 4 = RGB + depth (the headline "RGB+ED" frames), 3 = RGB
dev = "cuda"
if os.environ.get("SCENE", "") == "heavy":     # the heavy-tailed scene (not a BASELINE config): bench.py `heavy_tailed`
    from robosimgs_amd import synthetic_scene_heavy_tailed
    g = synthetic_scene_heavy_tailed(n, sh_degree=deg, seed=0)
else:
    g = synthetic_scene(n, math.log(mu), deg, 0)
if os.environ.get("MORTON", "1") != "0":      # the order FrameRenderer keeps its resident scene in
    g = g.sorted_by_locality()
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
def project(lean=False):
    # lean = the inference-frame form bench.py's frames run: splats + binning seed + depths only
    return ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, CH == 4, want_splats=True,
                                     bin_seed=("tight" if os.environ.get("TIGHT", "1") != "0" else "classic") if lean else None, lean=lean)
radii, m2d, dep, con, _, feats, splats = project()
TIGHT = os.environ.get("TIGHT", "1") != "0"      # tightened tile rectangles (the render path's default)
tkw = dict(conics=con, opacities=t["opacities"]) if TIGHT else {}
tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 40_000_000 if n > 2_000_000 else 8_000_000, want_tiles_per_gauss=False, want_pair_info=True, **tkw)
# the list capacity bench.py gives its frames (it decides which instantiation of the per-tile sort runs)
CAP = int(os.environ.get("CAP", int(int(tl.n_isect) * 1.25) + 4096))
# (SLOTS=0: the raster backward gathers the pairs' record slots from pair_info, as before round 5)
tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, CAP, want_tiles_per_gauss=False, want_pair_info=True,
                         splats=splats if os.environ.get("SLOTS", "1") != "0" else None, **tkw)
seed = project(lean=True)[-1]
out = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats)
torch.cuda.synchronize()
print("n_isect", int(tl.n_isect))
if os.environ.get("STATS"):
    ln = (tl.tile_offsets[1:] - tl.tile_offsets[:-1]).float()
    print("list length: mean %.0f median %.0f p99 %.0f max %d; tiles over 2048: %d, over 8192: %d, over 16384: %d" % (
        float(ln.mean()), float(ln.median()), float(torch.quantile(ln, 0.99)), int(ln.max()), int((ln > 2048).sum()), int((ln > 8192).sum()), int((ln > 16384).sum())))
vr = torch.rand(H, W, CH, device=dev); va = torch.rand(H, W, device=dev)
for _ in range(reps):
    if stage == "raster":
        ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, out=out, splats=splats, group_order=tl.group_order)
    elif stage == "raster_inf":          # the inference variant (no last_ids): the kernel bench.py times
        ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, out=(out[0], out[1], None), track_last=False, splats=splats, expected_last=CH == 4, group_order=tl.group_order)
    elif stage == "raster_inf_q":        # the same, one wave per 8x8 block (MGS_RASTER_LATENCY)
        ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, out=(out[0], out[1], None), track_last=False, splats=splats, expected_last=CH == 4, latency=True, group_order=tl.group_order)
    elif stage == "raster_bwd":
        ops.rasterize_bwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, out[1], out[2], vr, va)
    elif stage == "raster_bwd_det":
        ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl, out[1], out[2], vr, va, splats=splats)
    elif stage == "raster_bwd_split":    # segmented walk (forward checkpoints), SEG entries per unit
        SEG = int(os.environ.get("SEG", 256))
        if "ck" not in globals():
            ck = ops.checkpoint_buffer(CAP, tw, th, CH, SEG, dev)
            out_s = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats,
                                          latency=True, expected_last=CH == 4, group_order=tl.group_order, channels=CH, checkpoints=ck, checkpoint_interval=SEG)
        ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl, out_s[1], out_s[2], vr, va, splats=splats,
                                  expected_render=out_s[0] if CH == 4 else None, render_out=out_s[0], checkpoints=ck, checkpoint_interval=SEG)
    elif stage == "train":
        from robosimgs_amd.rendering import rasterization
        ps = {k: t[k].detach().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
        c, a_, _ = rasterization(ps["means"], ps["quats"], ps["scales"], ps["opacities"], ps["colors"], vm[None], K[None], W, H, sh_degree=deg, isect_capacity=CAP, tile_bounds="tight" if TIGHT else "classic")
        (c - vr[None, ..., :3]).abs().mean().backward()
    elif stage == "binning":
        # seeded by the projection kernel, no tile ids: what a frame runs (the seed's sums are scanned in place:
        # restore them first)
        seed = project(lean=True)[-1]
        ops.isect_tiles_raw(None, None, dep, tw, th, CAP, want_tiles_per_gauss=False, seed=seed, want_tile_ids=False)
    elif stage == "project":
        project(lean=True)
torch.cuda.synchronize()
