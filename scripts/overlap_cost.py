"""What each stage costs in THROUGHPUT terms with 3 frames in flight: replay graphs that contain
only some of the stages (on the tile lists of a real frame) and compare ms/frame."""
import math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops

dev = "cuda"
n, W, H, deg = 1_000_000, 1920, 1080, 3
g = synthetic_scene(n, math.log(0.012), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
cap = 4_700_000

def project():
    return ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H,
                                     0.3, 0.01, 1e10, 0.0, False, False, want_splats=True)
def binning(p):
    return ops.isect_tiles_raw(p[1], p[0], p[2], tw, th, cap, want_tiles_per_gauss=False, conics=p[3], opacities=t["opacities"])
def raster(p, tl, out=None):
    return ops.rasterize_fwd_raw(p[1], p[3], p[5], t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids,
                                 splats=p[6], track_last=False, out=out)

def run(stages, nslots=3, iters=300):
    slots = []
    for _ in range(nslots):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            p = project(); tl = binning(p); o = raster(p, tl)
            torch.cuda.synchronize()
            def body():
                pp = project() if "P" in stages else p
                tt = binning(pp) if "B" in stages else tl
                if "R" in stages: raster(pp, tt, out=o)
            body(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                body()
        slots.append((st, gr, p, tl, o))
    torch.cuda.synchronize()
    def go(k):
        for i in range(k):
            st, gr = slots[i % nslots][:2]
            with torch.cuda.stream(st):
                gr.replay()
    go(30); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(iters); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

for stages in ("R", "P", "B", "PR", "BR", "PB", "PBR"):
    print(f"stages {stages:4s} 1 slot {run(stages, 1):.3f} ms   3 slots {run(stages, 3):.3f} ms/frame")
