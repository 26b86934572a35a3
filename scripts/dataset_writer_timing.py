"""End to end data generation at 1080p: render (FrameRenderer, dataset frames out of the raster) -> device-to-host ->
PNG + npy.gz on disk (DatasetWriter), against the renderer alone.  The reference's only stated consumer of the render path
reads these files (Articulation/utils/nerf2physic_utils.py:84-118).
    python scripts/dataset_writer_timing.py [n_frames]"""
import math, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from robosimgs_amd import FrameRenderer, camera_ring, synthetic_scene
from robosimgs_amd.dataset import DatasetWriter, encode_npy_gz, encode_png_rgba

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 48
W, H = 1920, 1080
g = synthetic_scene(1_000_000, math.log(0.012), 3, 0)
t = g.to_torch("cuda", 3)
cams = camera_ring(n_frames, W, H)
fr = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=3, sizing_camera=(cams[0].viewmat(), cams[0].K), capacity_margin=1.6,
                   dataset_output=torch.float32, dataset_K=cams[0].K)
root = tempfile.mkdtemp(prefix="mgs_ds_", dir=os.environ.get("TMPDIR", "/tmp"))


def run(consume):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fr.render_sequence(cams, consume)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


run(lambda i, f: None)
dt = run(lambda i, f: None)
print(f"render only ({n_frames} frames of the 64-camera ring's kind, dataset frames out of the raster): {n_frames / dt:8.1f} frames/s")
host = {}
dt = run(lambda i, f: host.__setitem__(i % 4, (f["rgba"].cpu(), f["distance"].cpu())))
print(f"render + device-to-host (pageable, synchronous): {n_frames / dt:8.1f} frames/s")
rgba, dist = host[0][0].numpy(), host[0][1].numpy()
for lvl_p, lvl_g in ((6, 9), (1, 1)):
    t0 = time.perf_counter(); b1 = encode_png_rgba(rgba, lvl_p); t1 = time.perf_counter(); b2 = encode_npy_gz(dist, lvl_g); t2 = time.perf_counter()
    print(f"one thread, one frame: PNG level {lvl_p} {1e3 * (t1 - t0):6.1f} ms ({len(b1) / 1e6:.2f} MB), npy.gz level {lvl_g} {1e3 * (t2 - t1):6.1f} ms "
          f"({len(b2) / 1e6:.2f} MB of {dist.nbytes / 1e6:.2f})")
print("| DatasetWriter | PNG / gzip level | frames/s to disk |")
print("|---|---|---:|")
for workers, lvl_p, lvl_g in ((0, 6, 9), (8, 6, 9), (32, 6, 9), (64, 6, 9), (0, 1, 1), (32, 1, 1), (64, 1, 1)):
    d = os.path.join(root, f"w{workers}_{lvl_p}")
    n = n_frames if workers else max(6, n_frames // 8)
    with DatasetWriter(d, workers=workers, png_level=lvl_p, gz_level=lvl_g) as wr:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fr.render_sequence(cams[:n], lambda i, f: wr.write(i, f["rgba"], f["distance"]))
        wr.flush()
        dt = time.perf_counter() - t0
    print(f"| {'one thread (write() encodes)' if not workers else str(workers) + ' pool threads'} | {lvl_p} / {lvl_g} | {n / dt:.1f} |")
    shutil.rmtree(d)
shutil.rmtree(root, ignore_errors=True)
print("host:", os.cpu_count(), "logical cores")
