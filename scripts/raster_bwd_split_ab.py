"""Segmented backward raster (forward checkpoints) against the whole-list walk at config 2, 4 channels, RGB+ED:
HIP-event time of memset + raster_bwd + reduce per segment length, the forward's extra cost for the checkpoint
stores, and the largest difference between the gradients."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
n, mu, W, H, deg = 1_000_000, 0.012, 1920, 1080, 3
dev = "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0)
if os.environ.get("MORTON", "0") != "0":
    g = g.sorted_by_locality()
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
CAP = 4_700_000
radii, m2d, dep, con, _, feats, splats = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True)
tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, CAP, want_tiles_per_gauss=False, want_pair_info=True, conics=con, opacities=t["opacities"])
torch.manual_seed(0)
vr = torch.rand(H, W, 4, device=dev) - 0.5; va = torch.rand(H, W, device=dev) - 0.5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

SEGS = [0] + [int(x) for x in os.environ.get("SEGS", "64,128,256,512").split(",")]
var = {}
for seg in SEGS:
    ck = ops.checkpoint_buffer(CAP, tw, th, 4, seg, dev) if seg else None
    def fwd(ck=ck, seg=seg):
        return ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids,
                                     splats=splats, latency=True, expected_last=True, group_order=tl.group_order, channels=4,
                                     checkpoints=ck, checkpoint_interval=seg)
    out = fwd()
    def bwd(ck=ck, seg=seg, out=out):
        return ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl, out[1], out[2], vr, va,
                                         splats=splats, expected_render=out[0], render_out=out[0] if seg else None,
                                         checkpoints=ck, checkpoint_interval=seg)
    r, r2 = bwd(), bwd()
    var[seg] = dict(fwd=fwd, bwd=bwd, r=r, same=all(torch.equal(a, b) for a, b in zip(r[:4], r2[:4])), tf=[], tb=[])

def timed(fn, reps=10):
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

# the variants take turns (the clock drifts over a run: back-to-back blocks of one variant each would measure that)
for rnd in range(int(os.environ.get("ROUNDS", 7))):
    for seg in SEGS:
        var[seg]["tf"].append(timed(var[seg]["fwd"]))
        var[seg]["tb"].append(timed(var[seg]["bwd"]))
ref = var[0]["r"]
for seg in SEGS:
    v = var[seg]
    line = (f"segment {seg:4d}: forward (per block, last_ids) {np.median(v['tf']):7.1f} us   memset + raster_bwd + reduce "
            f"{np.median(v['tb']):7.1f} us (min {min(v['tb']):.1f})   reproducible {v['same']}")
    if seg:
        d = [float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30) for a, b in zip(v["r"][:4], ref[:4])]
        line += "   max |d| / max |ref| (means2d, conics, feats, opacities): " + " ".join(f"{x:.2e}" for x in d)
    print(line, flush=True)
