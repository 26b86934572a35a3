"""Experiment: raster on a LOW-priority stream, projection+binning on a normal one (two graphs per slot)."""
import ctypes, math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops
dev = "cuda"; torch.zeros(1, device=dev)
hip = ctypes.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])
lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print("priority range: least", lo.value, "greatest", hi.value)
def prio_stream(p):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, p) == 0   # hipStreamNonBlocking
    return torch.cuda.ExternalStream(s.value)
n, W, H, deg = 1_000_000, 1920, 1080, 3
g = synthetic_scene(n, math.log(0.012), deg, 0); cam = camera_ring(1, W, H, thetas=[0.3])[0]; t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev); K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16); cap = 4_700_000
def front():
    p = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, False, want_splats=True)
    return p, ops.isect_tiles_raw(p[1], p[0], p[2], tw, th, cap, want_tiles_per_gauss=False, conics=p[3], opacities=t["opacities"])
def back(f, out=None):
    p, tl = f
    return ops.rasterize_fwd_raw(p[1], p[3], p[5], t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=p[6], track_last=False, out=out)
def capture(stream, fn):
    with torch.cuda.stream(stream):
        for _ in range(2): r = fn()
        torch.cuda.synchronize(); gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream): r = fn()
    torch.cuda.synchronize(); return gr, r
def run(nslots, pf, pb):
    slots = []
    for _ in range(nslots):
        sf, sb = prio_stream(pf), prio_stream(pb)
        g1, f = capture(sf, front); g2, o = capture(sb, lambda: back(f))
        slots.append((sf, sb, g1, g2, torch.cuda.Event(), torch.cuda.Event()))
    torch.cuda.synchronize()
    def go(k):
        for i in range(k):
            sf, sb, g1, g2, e1, e2 = slots[i % nslots]
            with torch.cuda.stream(sf): sf.wait_event(e2); g1.replay(); e1.record(sf)
            with torch.cuda.stream(sb): sb.wait_event(e1); g2.replay(); e2.record(sb)
    go(30); torch.cuda.synchronize(); t0 = time.perf_counter(); go(300); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 300 * 1e3
for nslots in (2, 3):
    for pf, pb in ((0, 0), (0, lo.value), (hi.value, 0), (hi.value, lo.value)):
        ms = run(nslots, pf, pb)
        print(f"{nslots} slots, front priority {pf}, raster priority {pb}: {ms:.3f} ms/frame ({1e3/ms:.0f} frames/s)")
