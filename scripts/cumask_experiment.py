"""Experiment: partition the CUs between the raster (VALU-bound) and projection+binning
(HBM / latency-bound) with hipExtStreamCreateWithCUMask streams; two HIP graphs per slot."""
import ctypes, math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops

hip = ctypes.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]) if any("libamdhip64" in l for l in open("/proc/self/maps")) else None
dev = "cuda"
torch.zeros(1, device=dev)
hip = ctypes.CDLL([l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0])

def masked_stream(cu_lo, cu_hi, total=256):
    """stream restricted to CUs [cu_lo, cu_hi) of every ... (bit i of the mask = CU i)"""
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for i in range(cu_lo, cu_hi):
        mask[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

n, W, H, deg = 1_000_000, 1920, 1080, 3
g = synthetic_scene(n, math.log(0.012), deg, 0)
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
cap = 4_700_000

def front():
    p = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H,
                                  0.3, 0.01, 1e10, 0.0, False, False, want_splats=True)
    tl = ops.isect_tiles_raw(p[1], p[0], p[2], tw, th, cap, want_tiles_per_gauss=False, conics=p[3], opacities=t["opacities"])
    return p, tl

def back(f, out=None):
    p, tl = f
    return ops.rasterize_fwd_raw(p[1], p[3], p[5], t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids,
                                 splats=p[6], track_last=False, out=out)

def capture(stream, fn):
    with torch.cuda.stream(stream):
        for _ in range(2): r = fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            r = fn()
    torch.cuda.synchronize()
    return gr, r

def run(nslots, front_cus, interleave):
    slots = []
    for _ in range(nslots):
        if front_cus == 0:
            sf, sb = torch.cuda.Stream(), torch.cuda.Stream()
        elif interleave:   # every 256/front_cus-th CU to the front (spread over all XCDs)
            step = 256 // front_cus
            words = 8
            mf = (ctypes.c_uint32 * words)(); mb = (ctypes.c_uint32 * words)()
            for i in range(256):
                (mf if i % step == 0 else mb)[i // 32] |= 1 << (i % 32)
            a = ctypes.c_void_p(); b = ctypes.c_void_p()
            assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(a), words, mf) == 0
            assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(b), words, mb) == 0
            sf, sb = torch.cuda.ExternalStream(a.value), torch.cuda.ExternalStream(b.value)
        else:
            sf, sb = masked_stream(0, front_cus), masked_stream(front_cus, 256)
        g1, f = capture(sf, front)
        g2, o = capture(sb, lambda: back(f))
        slots.append((sf, sb, g1, g2, torch.cuda.Event(), torch.cuda.Event()))
    torch.cuda.synchronize()
    def go(iters):
        for i in range(iters):
            sf, sb, g1, g2, e1, e2 = slots[i % nslots]
            with torch.cuda.stream(sf):
                sf.wait_event(e2); g1.replay(); e1.record(sf)
            with torch.cuda.stream(sb):
                sb.wait_event(e1); g2.replay(); e2.record(sb)
    go(30); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(300); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 300 * 1e3

for nslots in (3, 4):
    print(f"{nslots} slots, no masks: {run(nslots, 0, False):.3f} ms/frame")
    for fc in (32, 64):
        for inter in (False, True):
            ms = run(nslots, fc, inter)
            print(f"{nslots} slots, front on {fc} CUs ({'interleaved' if inter else 'contiguous'}), raster on {256 - fc}: {ms:.3f} ms/frame ({1e3 / ms:.0f} frames/s)")
