"""Per-tile timeline of the backward raster (measurement build: MGS_RASTER_BWD_FLAGS=-DMGS_RASTER_BWD_TIMING).
Every wave stamps its start and end on the 100 MHz clock with the entries it walked and the pairs it evaluated;
this prints where the launch's time goes: the span, the resident waves over time, the cost per pair of waves that
run crowded (early) and alone (late), and the tiles that end last."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, _lib
n, mu, W, H, deg = 1_000_000, 0.012, 1920, 1080, 3
dev = "cuda"
g = synthetic_scene(n, math.log(mu), deg, 0)
if os.environ.get("MORTON", "1") != "0":
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
for _ in range(3):
    ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl, out[1], out[2], vr, va, splats=splats)
torch.cuda.synchronize()
nt = tw * th
buf = (ctypes.c_ulonglong * (4 * nt))()
fn = _lib.lib().mgs_debug_bwd_times
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, 4 * nt) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nt, 4).astype(np.int64)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
beg, end, walked, pairs = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2], a[:, 3]     # us
span = end.max()
dur = end - beg
print(f"tiles with work {len(a)}  span {span:.1f} us  sum of wave time {dur.sum() / 1e3:.1f} ms = {dur.sum() / span:.0f} waves resident on average ({dur.sum() / span / 1024:.2f} per SIMD)")
print(f"pairs {pairs.sum()}  walked {walked.sum()}  ns per pair per wave: all {1e3 * dur.sum() / pairs.sum():.1f}")
for lo, hi in ((0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
    edges = np.linspace(lo * span, hi * span, 2)
    res = ((beg < hi * span) & (end > lo * span))
    # resident waves at the midpoint of the window
    mid = 0.5 * (lo + hi) * span
    r = ((beg <= mid) & (end > mid)).sum()
    print(f"  at {mid:6.1f} us: {r} waves resident ({r / 1024:.2f} per SIMD)")
order = np.argsort(-end)
print("last to end:   end us   start us   dur us   walked  pairs  ns/pair")
for i in order[:12]:
    print(f"              {end[i]:7.1f}  {beg[i]:7.1f}  {dur[i]:7.1f}  {walked[i]:6d} {pairs[i]:6d}  {1e3 * dur[i] / max(1, pairs[i]):.0f}")
big = np.argsort(-pairs)[:12]
print("most pairs:    end us   start us   dur us   walked  pairs  ns/pair")
for i in big:
    print(f"              {end[i]:7.1f}  {beg[i]:7.1f}  {dur[i]:7.1f}  {walked[i]:6d} {pairs[i]:6d}  {1e3 * dur[i] / max(1, pairs[i]):.0f}")
# cost per pair against how crowded the chip was: waves that start in the first tenth vs the last tenth
early, late = beg < 0.1 * span, beg > 0.6 * span
for name, m in (("start < 10 % of span", early), ("start > 60 % of span", late)):
    if m.any():
        print(f"  {name}: {m.sum()} tiles, {1e3 * dur[m].sum() / max(1, pairs[m].sum()):.0f} ns per pair, {1e3 * dur[m].sum() / max(1, walked[m].sum()):.0f} ns per walked entry")
hist, e = np.histogram(pairs, bins=[0, 100, 200, 400, 800, 1200, 1600, 2400, 3200, 100000])
print("pairs per tile histogram:", list(zip(e[:-1].tolist(), hist.tolist())))
