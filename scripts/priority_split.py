"""Experiment: with three frames in flight, a frame's projection + binning (latency-bound, few waves) as a graph of its own on
a HIGH-PRIORITY stream, the raster as a second graph on the slot's stream -- against FrameRenderer's one graph per frame.
The hardware queue of a high-priority stream is served first when the dispatcher has room: do the thin kernels stop
waiting behind other frames' raster workgroups?     python scripts/priority_split.py [prio]   (prio: -1 high, 0 = same)
"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, FrameRenderer
from robosimgs_amd.pipeline import independent_streams

PRIO = int(sys.argv[1]) if len(sys.argv) > 1 else -1
n, mu, W, H, deg, NFL = 1_000_000, 0.012, 1920, 1080, 3, 3
dev = torch.device("cuda")
g = synthetic_scene(n, math.log(mu), deg, 0).sorted_by_locality()
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
CAP = 4_700_000
print("stream priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")

fr = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=NFL, isect_capacity=CAP, reorder=None)
cam_dev = FrameRenderer.pack_camera(vm, K)

def stage_a():
    radii, m2d, dep, con, _, feats, splats, seed = ops.project_color_fwd_raw(
        t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True,
        want_splats=True, bin_seed="tight", lean=True)
    tl = ops.isect_tiles_raw(None, None, dep, tw, th, CAP, want_tiles_per_gauss=False, seed=seed, want_tile_ids=False)
    return splats, tl

def stage_b(splats, tl):
    return ops.rasterize_fwd_raw(None, None, None, None, None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, track_last=False,
                                 splats=splats, expected_last=True, latency=False, group_order=tl.group_order, channels=4)

slots = []
streams = independent_streams(dev, NFL)
for i in range(NFL):
    hp = torch.cuda.Stream(dev, priority=PRIO)
    s = streams[i]
    with torch.cuda.stream(hp):
        for _ in range(2): a = stage_a()
        torch.cuda.synchronize()
        ga = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, stream=hp):
            a = stage_a()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(2): b = stage_b(*a)
        torch.cuda.synchronize()
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb, stream=s):
            b = stage_b(*a)
    torch.cuda.synchronize()
    slots.append(dict(hp=hp, s=s, ga=ga, gb=gb, a=a, b=b, mid=torch.cuda.Event(), done=torch.cuda.Event(), released=torch.cuda.Event()))

# same pixels as the renderer's frame
tk = fr.submit(cam_dev); f = fr.fetch(tk, check=False)
sl = slots[0]
with torch.cuda.stream(sl["hp"]): sl["ga"].replay(); sl["mid"].record(sl["hp"])
with torch.cuda.stream(sl["s"]): sl["s"].wait_event(sl["mid"]); sl["gb"].replay()
torch.cuda.synchronize()
print("identical frame:", torch.equal(f["colors"].reshape(-1), sl["b"][0].reshape(-1)))
fr.release(tk)

def fps_renderer(frames=200):
    tickets = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(frames):
        if len(tickets) == NFL:
            k = tickets.pop(0); fr.fetch(k, check=False); fr.release(k)
        tickets.append(fr.submit(cam_dev))
    while tickets:
        k = tickets.pop(0); fr.fetch(k, check=False); fr.release(k)
    torch.cuda.synchronize()
    return frames / (time.perf_counter() - t0)

def fps_split(frames=200):
    cur = torch.cuda.current_stream(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(frames):
        sl = slots[i % NFL]
        if i >= NFL:                       # the consumer "fetches" the slot's previous frame, then the slot is free
            cur.wait_event(sl["done"]); sl["released"].record(cur)
            sl["hp"].wait_event(sl["released"])
        with torch.cuda.stream(sl["hp"]):
            sl["ga"].replay(); sl["mid"].record(sl["hp"])
        with torch.cuda.stream(sl["s"]):
            sl["s"].wait_event(sl["mid"]); sl["gb"].replay(); sl["done"].record(sl["s"])
    torch.cuda.synchronize()
    return frames / (time.perf_counter() - t0)

res = {"renderer": [], "split": []}
for rnd in range(7):
    res["renderer"].append(fps_renderer()); res["split"].append(fps_split())
for k, v in res.items():
    print(f"{k}: {np.median(v[1:]):.0f} frames/s (median of {len(v) - 1} interleaved rounds of 200 frames; all: {' '.join('%.0f' % x for x in v)})")
