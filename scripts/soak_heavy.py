"""Soak (not part of pytest): random CLUSTERED, heavy-tailed scenes (synthetic_scene_heavy_tailed with random sizes, cluster
counts, needle / screen-filling shares, resolutions and cameras) through what the long lists exercise:
  * the tile lists against the stable-sort formulation on the GPU's own projected inputs, bit for bit (lists of thousands of
    entries, buckets of near-identical depths: the per-tile sort's long-list launch and its generic path);
  * classic against tightened rectangles: same image, same gradients (whole-list walk), bit for bit;
  * the gates of tests/test_gpu_heavy.py on the seed's scene: blend stage absolute, projection stage, whole path relative;
  * gradients finite and bit-reproducible.
    python scripts/soak_heavy.py [n_scenes [first_seed]]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cpu_ref
from oracle import gs_oracle_np as O
from robosimgs_amd import camera_ring, ops, rasterization, synthetic_scene_heavy_tailed
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_heavy as heavy_gates
DEV = "cuda"
def _t(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad, longest, worst = 0, 0, 0.0
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # python scripts/soak_heavy.py n_scenes [first_seed]
for seed in range(first, first + n_scenes):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(60_000, 400_000)); W = int(rng.integers(300, 1300)); H = int(rng.integers(200, 800)); deg = int(rng.integers(0, 4))
    g = synthetic_scene_heavy_tailed(n, math.log(float(rng.uniform(0.004, 0.03))), deg, seed, n_clusters=int(rng.integers(3, 120)),
                                     n_screen_filling=int(rng.integers(0, 9)), n_needles=int(rng.integers(0, n // 20)))
    cam = camera_ring(1, W, H, thetas=[float(rng.uniform(0, 6.28))], radius=float(rng.uniform(4, 9)))[0]
    t = g.to_torch(DEV, deg); vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    tw, th = -(-W // 16), -(-H // 16)
    # lists, bit for bit (the operator path: classic rectangles, read-back capacity)
    radii, m2d, dep, con, _ = ops.fully_fused_projection(t["means"], None, t["quats"], t["scales"], vm, K, W, H)
    tpg, ids, flat = ops.isect_tiles(m2d, radii, dep, 16, tw, th)
    keys_ref, order = torch.sort(ids, stable=True)
    ok_lists = bool((ids[1:] >= ids[:-1]).all())
    tiles = (ids >> 32); same = ids[1:] == ids[:-1]
    ok_lists = ok_lists and bool((flat.long()[1:][same] > flat.long()[:-1][same]).all())
    ok_lists = ok_lists and bool(((ids & 0xffffffff) == dep[0][flat.long()].view(torch.int32).long()).all())
    ok_lists = ok_lists and bool((torch.bincount(flat.long(), minlength=n) == tpg[0].long()).all())
    lens = torch.bincount(tiles, minlength=tw * th)
    longest = max(longest, int(lens.max()))
    outs = []
    for b in ("classic", "tight"):
        p = {k: t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
        c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, W, H, sh_degree=deg, render_mode="RGB+ED",
                                   tile_bounds=b, backward_segment=0)
        (c.sum() + a.sum()).backward()
        outs.append((c.detach(), a.detach(), [v.grad for v in p.values()]))
    same_img = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and all(torch.equal(x, y) for x, y in zip(outs[0][2], outs[1][2]))
    fin = all(bool(torch.isfinite(x).all()) for x in outs[1][2])
    # segmented walk: bit-reproducible
    gr = []
    for _ in range(2):
        p = {k: t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
        c, a, _m = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, W, H, sh_degree=deg, render_mode="RGB+ED")
        (c.sum() + a.sum()).backward()
        gr.append([v.grad for v in p.values()])
    repro = all(torch.equal(x, y) for x, y in zip(*gr))
    # the gates of tests/test_gpu_heavy.py (round 6): the blend stage on the GPU's own inputs ABSOLUTE, the projection stage, the whole
    # path relative to the port's fp32 instantiation (percentiles and far outliers; no count clauses on these small frames)
    try:
        heavy_gates.test_soak_scene_lists_stage_blend_and_whole_path(seed)
        gate = True
    except AssertionError as e:
        gate = False
        print("GATE", str(e)[:600])
    if not (ok_lists and same_img and fin and repro and gate):
        bad += 1
        print("FAIL seed", seed, n, W, H, deg, dict(lists=ok_lists, tight=same_img, finite=fin, reproducible=repro, gate=gate))
print(f"{n_scenes} heavy-tailed scenes (seeds {first} .. {first + n_scenes - 1}): failures {bad}; longest tile list {longest} entries")
