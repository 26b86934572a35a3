"""Same-process, interleaved A/B of two BUILDS of libmgs.so for the BACKWARD raster (the shipped one against a variant of
raster_bwd.hip compiled with extra flags): memset + unit tables + raster backward + reduce between HIP events, the two
builds taking turns; gradients compared bit for bit.
    python scripts/ab_builds_bwd.py "-DMGS_RASTER_BWD_IDS_AHEAD=1"
"""
import math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, _lib
from robosimgs_amd.csrc import build as B

flags = sys.argv[1] if len(sys.argv) > 1 else ""
src = sys.argv[2] if len(sys.argv) > 2 else "raster_bwd.hip"
B.build()
objs = {s: os.path.join(B.OBJ_DIR, s.replace(".hip", ".o")) for s in B.SOURCES}
obj = os.path.join(B.OBJ_DIR, src.replace(".hip", ".variant.o"))
subprocess.run([B._hipcc(), *B.FLAGS, *B.PER_SOURCE_FLAGS.get(src, []), *flags.split(), "-c", os.path.join(B.HERE, src), "-o", obj], check=True)
objs[src] = obj
VAR = os.path.join(B.HERE, "libmgs_variant.so")
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs.values(), "-o", VAR], check=True)
libs = {"A (shipped)": _lib._load(), "B (variant)": _lib._load(VAR)}

n, mu, W, H, deg, SEG = 1_000_000, 0.012, 1920, 1080, 3, 256
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
state = {}
for name, L in libs.items():
    _lib._lib = L
    radii, m2d, dep, con, _, feats, splats, seed = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight")
    tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, CAP, want_tiles_per_gauss=False, want_pair_info=True, seed=seed, splats=splats)
    ck = ops.checkpoint_buffer(CAP, tw, th, 4, SEG, dev)
    render, alphas, last = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats,
                                                 expected_last=True, latency=True, group_order=tl.group_order, channels=4, checkpoints=ck, checkpoint_interval=SEG)
    v_r = torch.sign(render - torch.rand(H, W, 4, device=dev, generator=torch.Generator(dev).manual_seed(1))) / float(render.numel())
    def run(only, st=None, m2d=m2d, con=con, feats=feats, splats=splats, tl=tl, alphas=alphas, last=last, v_r=v_r, render=render, ck=ck):
        return ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl, alphas, last, v_r, None, splats=splats,
                                         expected_render=render, render_out=render, checkpoints=ck, checkpoint_interval=SEG, records_only=only)
    state[name] = dict(L=L, run=run, out=run(False), t={"records": [], "all": []})
a, b = state.values()
print("identical gradients:", all(torch.equal(x, y) for x, y in zip(a["out"][:4], b["out"][:4])))
def timed(fn, reps):
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rnd in range(int(os.environ.get("ROUNDS", 7))):
    for name, st in state.items():
        _lib._lib = st["L"]
        st["t"]["records"].append(timed(lambda: st["run"](True), 10))
        st["t"]["all"].append(timed(lambda: st["run"](False), 10))
for name, st in state.items():
    tt = {k: float(np.median(v[1:])) for k, v in st["t"].items()}
    print(f"{name}: memset + unit tables + raster backward {tt['records']:.1f} us, + reduce {tt['all']:.1f} us  (medians of {len(st['t']['all']) - 1} interleaved rounds of 10, back to back)")
