"""Per-unit timeline of the backward raster (measurement build: MGS_RASTER_BWD_FLAGS=-DMGS_RASTER_BWD_TIMING).
Every wave stamps enter / walk begins / end on the 100 MHz clock with the entries it walked and the pairs it evaluated;
SEG=0 whole-list walk (a unit is a tile), SEG=128 ... the segmented walk."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, _lib
n, mu, W, H, deg = 1_000_000, 0.012, 1920, 1080, 3
SEG = int(os.environ.get("SEG", 0))
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
ck = ops.checkpoint_buffer(CAP, tw, th, 4, SEG, dev) if SEG else None
out = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats, latency=True,
                            channels=4, checkpoints=ck, checkpoint_interval=SEG)
torch.manual_seed(0)
vr = torch.rand(H, W, 4, device=dev); va = torch.rand(H, W, device=dev)
NW = 5 * 65536
buf = (ctypes.c_ulonglong * NW)()
fn = _lib.lib().mgs_debug_bwd_times
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3):
    ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl, out[1], out[2], vr, va, splats=splats,
                              render_out=out[0] if SEG else None, checkpoints=ck, checkpoint_interval=SEG)
    torch.cuda.synchronize()
    assert fn(buf, NW) == 0           # (reads and clears: the last launch's stamps are the ones analysed)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 5).astype(np.int64)
a = a[a[:, 2] > 0]
t0 = a[:, 0].min()
ent, beg, end, walked, pairs = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0, a[:, 3], a[:, 4]     # us
span = end.max()
dur = end - ent
print(f"SEG {SEG}: units with work {len(a)}  span {span:.1f} us  sum of wave time {dur.sum() / 1e3:.1f} ms = {dur.sum() / span:.0f} waves resident on average ({dur.sum() / span / 1024:.2f} per SIMD)")
print(f"pairs {pairs.sum()}  walked {walked.sum()}  ns per pair per wave {1e3 * dur.sum() / pairs.sum():.1f};  prologue (enter -> walk) mean {np.mean(beg - ent):.2f} us, p90 {np.quantile(beg - ent, 0.9):.2f}, sum {np.sum(beg - ent) / 1e3:.1f} ms")
for f in (0.05, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 0.95):
    mid = f * span
    r = ((ent <= mid) & (end > mid)).sum()
    st = ((ent > mid - 5) & (ent <= mid)).sum()
    print(f"  at {mid:6.1f} us: {r} units resident ({r / 1024:.2f} per SIMD); {st} entered in the 5 us before")
order = np.argsort(-end)
print("last to end:   end us   enter us   dur us   walked  pairs  ns/pair")
for i in order[:8]:
    print(f"              {end[i]:7.1f}  {ent[i]:7.1f}  {dur[i]:7.1f}  {walked[i]:6d} {pairs[i]:6d}  {1e3 * dur[i] / max(1, pairs[i]):.0f}")
early, late = ent < 0.1 * span, ent > 0.6 * span
for name, m in (("enter < 10 % of span", early), ("enter > 60 % of span", late)):
    if m.any():
        print(f"  {name}: {m.sum()} units, {1e3 * dur[m].sum() / max(1, pairs[m].sum()):.0f} ns per pair, {1e3 * dur[m].sum() / max(1, walked[m].sum()):.0f} ns per walked entry, mean duration {dur[m].mean():.1f} us")
hist, e = np.histogram(pairs, bins=[0, 25, 50, 100, 200, 400, 800, 100000])
print("pairs per unit histogram:", list(zip(e[:-1].tolist(), hist.tolist())))
hist, e = np.histogram(dur, bins=[0, 10, 25, 50, 100, 200, 400, 100000])
print("duration (us) histogram:", list(zip(e[:-1].tolist(), hist.tolist())))
