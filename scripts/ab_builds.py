"""Same-process, interleaved A/B of two BUILDS of libmgs.so (the shipped one against a variant of one source compiled
with extra flags): kernel-alone times of the forward stages and frames/s with three frames in flight, the two builds
taking turns so that clock drift hits both alike.
    python scripts/ab_builds.py raster_fwd.hip "-DMGS_RASTER_CLOSE_BRANCH=1" [tile_sort.hip "-D..."]
VARIANT_LIB=<path of a libmgs.so built from another tree>: that library is B (nothing is compiled); SCENE=4k | heavy.
"""
import math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, _lib, FrameRenderer
from robosimgs_amd.csrc import build as B

pairs = list(zip(sys.argv[1::2], sys.argv[2::2]))
B.build()
objs = {s: os.path.join(B.OBJ_DIR, s.replace(".hip", ".o")) for s in B.SOURCES}
for src, fl in pairs:
    obj = os.path.join(B.OBJ_DIR, src.replace(".hip", ".variant.o"))
    subprocess.run([B._hipcc(), *B.FLAGS, *B.PER_SOURCE_FLAGS.get(src, []), *fl.split(), "-c", os.path.join(B.HERE, src), "-o", obj], check=True)
    objs[src] = obj
VAR = os.path.join(B.HERE, "libmgs_variant.so")
if os.environ.get("VARIANT_LIB"):
    VAR = os.path.abspath(os.environ["VARIANT_LIB"])
else:
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs.values(), "-o", VAR], check=True)
libs = {"A (shipped)": _lib._load(), "B (variant)": _lib._load(VAR)}

# (SCENE=4k: configs[4] -- 5 M Gaussians at 3840x2160, the capacity bench.py's leg uses there)
if os.environ.get("SCENE") == "4k":
    n, mu, W, H, deg = 5_000_000, 0.008, 3840, 2160, 3
else:
    n, mu, W, H, deg = 1_000_000, 0.012, 1920, 1080, 3
dev = "cuda"
if os.environ.get("SCENE") == "heavy":
    from robosimgs_amd import synthetic_scene_heavy_tailed
    g = synthetic_scene_heavy_tailed(n, sh_degree=deg, seed=0)
else:
    g = synthetic_scene(n, math.log(mu), deg, 0)
if os.environ.get("MORTON", "1") != "0":
    g = g.sorted_by_locality()
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch(dev, deg)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
tw, th = -(-W // 16), -(-H // 16)
CAP = int(os.environ.get("CAP", 30_100_000 if os.environ.get("SCENE") == "4k" else (6_400_000 if os.environ.get("SCENE") == "heavy" else 4_700_000)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def timed(fn, reps):
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

state = {}
for name, L in libs.items():
    _lib._lib = L
    radii, m2d, dep, con, _, feats, splats, seed = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight")
    tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, CAP, want_tiles_per_gauss=False, seed=seed, want_tile_ids=False)
    st = dict(L=L, m2d=m2d, con=con, feats=feats, splats=splats, tl=tl, dep=dep)
    def proj(st=st):
        return ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight", lean=True)
    def binning(st=st):
        sd = proj()[-1]
        return ops.isect_tiles_raw(None, None, st["dep"], tw, th, CAP, want_tiles_per_gauss=False, seed=sd, want_tile_ids=False)
    def raster(lat, st=st):
        return ops.rasterize_fwd_raw(None, None, None, None, None, W, H, tw, th, st["tl"].tile_offsets, st["tl"].flatten_ids, track_last=False,
                                     splats=st["splats"], expected_last=True, latency=lat, group_order=st["tl"].group_order, channels=4)
    st.update(proj=proj, binning=binning, tile=lambda: raster(False), block=lambda: raster(True), out=raster(False),
              lists=(tl.flatten_ids.clone(), tl.tile_offsets.clone()), t={k: [] for k in ("proj", "proj+binning", "tile", "block", "fps")})
    st["fr"] = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=3, isect_capacity=CAP, reorder=None)
    st["cam"] = FrameRenderer.pack_camera(vm, K)
    state[name] = st
a, b = state.values()
n_is = int(a["tl"].n_isect)
print("identical lists:", torch.equal(a["lists"][0][:n_is], b["lists"][0][:n_is]) and torch.equal(a["lists"][1], b["lists"][1]),
      " identical frames:", torch.equal(a["out"][0], b["out"][0]) and torch.equal(a["out"][1], b["out"][1]))

def fps(st, frames=60 if os.environ.get("SCENE") == "4k" else 150):
    fr, tickets = st["fr"], []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(frames):
        if len(tickets) == 3:
            tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
        tickets.append(fr.submit(st["cam"]))
    while tickets:
        tk = tickets.pop(0); fr.fetch(tk, check=False); fr.release(tk)
    torch.cuda.synchronize()
    return frames / (time.perf_counter() - t0)

for rnd in range(int(os.environ.get("ROUNDS", 6))):
    for name, st in state.items():
        _lib._lib = st["L"]
        st["t"]["proj"].append(timed(st["proj"], 20))
        st["t"]["proj+binning"].append(timed(st["binning"], 20))
        st["t"]["tile"].append(timed(st["tile"], 20))
        st["t"]["block"].append(timed(st["block"], 20))
        st["t"]["fps"].append(fps(st))
for name, st in state.items():
    tt = {k: float(np.median(v[1:])) for k, v in st["t"].items()}
    print(f"{name}: projection {tt['proj']:.1f} us, binning {tt['proj+binning'] - tt['proj']:.1f} us, raster per tile {tt['tile']:.1f} us, per block "
          f"{tt['block']:.1f} us, 3 frames in flight {tt['fps']:.0f} frames/s  (medians of {len(st['t']['tile']) - 1} interleaved rounds)")
