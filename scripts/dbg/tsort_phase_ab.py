"""Where the per-tile sort's main kernel spends its time, round-5 tree against this one (configs[4] by default): libraries built
with -DMGS_TSORT_STOP=k (the kernel leaves after phase k: 1 group filter, 2 keys + highest differing bit, 3 histogram + scan,
4 scatter; the full kernel adds rank + store), binning stage timed with HIP events, the libraries taking turns.
    (here)  see profiles/r6/00_experiments.md section 1 for how the libmgs_{base,new}_stop{k}.so files are built
    gpurun -- 'SCENE=4k python scripts/dbg/tsort_phase_ab.py'"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from robosimgs_amd import synthetic_scene, camera_ring, ops, _lib
C = os.path.join(ROOT, "robosimgs_amd", "csrc")
names = ["base", "new"] + [f"{w}_stop{k}" for k in (1, 2, 3, 4) for w in ("base", "new")]
paths = {"base": os.path.join(C, "libmgs_base.so"), "new": os.path.join(C, "libmgs.so")}
paths.update({n: os.path.join(C, f"libmgs_{n}.so") for n in names if "stop" in n})
libs = {n: _lib._load(p) for n, p in paths.items() if os.path.exists(p)}
if os.environ.get("SCENE", "4k") == "4k":
    n, mu, W, H, CAP = 5_000_000, 0.008, 3840, 2160, 30_100_000
else:
    n, mu, W, H, CAP = 1_000_000, 0.012, 1920, 1080, 4_700_000
g = synthetic_scene(n, math.log(mu), 3, 0).sorted_by_locality()
cam = camera_ring(1, W, H, thetas=[0.3])[0]
t = g.to_torch("cuda", 3)
vm = torch.from_numpy(cam.viewmat().astype(np.float32)).cuda(); K = torch.from_numpy(cam.K.astype(np.float32)).cuda()
tw, th = -(-W // 16), -(-H // 16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {k: [] for k in libs}
for rnd in range(6):
    for name, L in libs.items():
        _lib._lib = L
        def proj():
            return ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], 3, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight", lean=True)
        seeds = [proj()[-1] for _ in range(10)]
        dep = proj()[2]
        torch.cuda.synchronize()
        e0.record()
        for sd in seeds:
            ops.isect_tiles_raw(None, None, dep, tw, th, CAP, want_tiles_per_gauss=False, seed=sd, want_tile_ids=False)
        e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
for name in libs:
    print(f"{name:12s} binning stage {np.median(res[name][1:]):7.1f} us")
