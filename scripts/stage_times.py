"""Per-stage timing of one forward frame (dev tool; torch events on the current stream)."""
import argparse, math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--mu", type=float, default=0.012)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--deg", type=int, default=3)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--classic", action="store_true", help="classic mean +- radius tile rectangles")
a = ap.parse_args()
dev = "cuda"
g = synthetic_scene(a.n, math.log(a.mu), a.deg, 0)
cam = camera_ring(1, a.w, a.h, thetas=[0.3])[0]
t = g.to_torch(dev, a.deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-a.w // 16), -(-a.h // 16)

def frame(cap, rec=None):
    def ev():
        e = torch.cuda.Event(enable_timing=True); e.record(); return e
    e0 = ev()
    radii, means2d, depths, conics, opac, feats, splats = ops.project_color_fwd_raw(
        t["means"], t["quats"], t["scales"], t["opacities"], a.deg, t["colors"], vm, K, a.w, a.h,
        0.3, 0.01, 1e10, 0.0, False, False, want_splats=True)
    e1 = ev()
    tkw = {} if a.classic else dict(conics=conics, opacities=t["opacities"])
    tl = ops.isect_tiles_raw(means2d, radii, depths, tw, th, cap, want_tiles_per_gauss=False, **tkw)
    e2 = ev()
    out = ops.rasterize_fwd_raw(means2d, conics, feats, t["opacities"], None, a.w, a.h, tw, th,
                                tl.tile_offsets, tl.flatten_ids, splats=splats, track_last=False)
    e3 = ev()
    if rec is not None:
        rec.append((e0, e1, e2, e3))
    return radii, tl, out

radii, tl, out = frame(40_000_000)
torch.cuda.synchronize()
n_isect = int(tl.n_isect.item())
print("n_vis", int((radii > 0).sum()), "n_isect", n_isect, "alpha mean", float(out[1].mean()))
cap = int(n_isect * 1.2)
rec = []
for _ in range(3): frame(cap)
for _ in range(a.iters): frame(cap, rec)
torch.cuda.synchronize()
ts = np.array([[x[i].elapsed_time(x[i + 1]) for i in range(3)] for x in rec])
print("ms median  project+SH %.3f  binning %.3f  raster %.3f  total %.3f" % (*np.median(ts, 0), np.median(ts.sum(1))))
# whole frame under a HIP graph
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): frame(cap)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        frame(cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters): gr.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
print("graph replay: %.3f ms/frame  (%.1f frames/s)" % (dt * 1e3, 1 / dt))

# two frames in flight: independent graphs (own buffers) replayed on two streams, so the
# latency-bound binning kernels of one frame overlap the VALU-bound raster of the other
for nstreams in (2, 3):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    graphs = []
    for st in streams:
        with torch.cuda.stream(st):
            for _ in range(2): frame(cap)
            torch.cuda.synchronize()
            gr2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr2, stream=st):
                keep = frame(cap)
            graphs.append((gr2, keep))
    torch.cuda.synchronize()
    iters = a.iters * 4
    t0 = time.perf_counter()
    for i in range(iters):
        st = streams[i % nstreams]
        with torch.cuda.stream(st):
            graphs[i % nstreams][0].replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("%d frames in flight: %.3f ms/frame  (%.1f frames/s)" % (nstreams, dt * 1e3, 1 / dt))
