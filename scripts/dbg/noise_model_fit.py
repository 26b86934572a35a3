"""How large is the kernel's error of alpha per pair, in units of the two magnitudes a noise model can use?  For soak seeds whose
blend-stage gate finds a pixel over 1e-4 + noise: at that pixel, every contributor's |d ln alpha| (the kernel's polynomial
about the tile centre, emulated in NumPy fp32 -- scripts/dbg/soak_pixel_cause.py:walk reproduces the HIP pixel -- against fp64)
over S (terms of sigma about the PIXEL) and over S_c (all monomials of the polynomial about the TILE CENTRE).
    python scripts/dbg/noise_model_fit.py 59 41 74"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts", "dbg"))
import numpy as np, torch
from oracle import cpu_ref, gs_oracle_np as O
import test_gpu_heavy as T
from robosimgs_amd import ops
f32 = np.float32
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))
LOG2E = f32(1.4426950408889634)
for seed in [int(a) for a in sys.argv[1:]] or [59]:
    g, cam, W, H, deg = T._soak_scene(seed)
    t, radii, m2d, dep, con, feats, splats, tl, tw, th = T._stage(g, cam, W, H, deg)
    r, a, _ = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids, splats=splats, group_order=tl.group_order)
    got, ga = r.cpu().numpy(), a.cpu().numpy()
    M2, CO, OP, FE = m2d.cpu().numpy(), con.cpu().numpy(), t["opacities"].cpu().numpy(), feats.cpu().numpy()
    ni = int(tl.n_isect); ids, offs = tl.flatten_ids[:ni].cpu().numpy(), tl.tile_offsets.cpu().numpy()
    ref, ra, info = cpu_ref.blend_f64(M2, CO, OP, FE, ids, offs, W, H, flip_eps=O.EPS_STAGE)
    ex = O.explained_pixels(info["margins"], O.EPS_STAGE, None)
    fm = info["feat_max"]
    lim = 1e-4 + 1.5 * info["noise_weight"][..., None] * 2 * fm[None, None, :]
    ratio = np.where(ex[..., None], 0, np.abs(got - ref) / lim).max(-1)
    py, px = np.unravel_index(np.argmax(ratio), ratio.shape)
    tile = (py // 16) * tw + px // 16
    lst = ids[offs[tile]:offs[tile + 1]]
    cx, cy = f32((px // 16) * 16 + 8), f32((py // 16) * 16 + 8)
    x, y = f32(px + 0.5) - cx, f32(py + 0.5) - cy
    print(f"##### seed {seed}: worst non-flip pixel ({py}, {px}) error / limit {ratio[py, px]:.2f}, |d| {np.abs(got - ref)[py, px]}, noise weight {info['noise_weight'][py, px]:.2e}, list {len(lst)}")
    T_, rows = 1.0, []
    for gid in lst:
        a_, b_, c_, o = CO[gid, 0], CO[gid, 1], CO[gid, 2], OP[gid]
        dx, dy = np.float64(M2[gid, 0]) - (px + 0.5), np.float64(M2[gid, 1]) - (py + 0.5)
        sg = 0.5 * (a_ * dx * dx + c_ * dy * dy) + b_ * dx * dy
        al = min(0.999, o * np.exp(-sg))
        if sg < 0 or al < 1 / 255:
            continue
        A, B, Cc, L = f32(f32(-0.5) * LOG2E * a_), f32(-LOG2E * b_), f32(f32(-0.5) * LOG2E * c_), f32(np.log2(np.float64(o)))
        mx, my = f32(M2[gid, 0] - cx), f32(M2[gid, 1] - cy)
        q0 = fma(mx, fma(B, my, f32(A * mx)), fma(f32(Cc * my), my, L))
        q1 = -fma(f32(2) * A, mx, f32(B * my)); q2 = -fma(f32(2) * Cc, my, f32(B * mx))
        pw = fma(Cc, f32(y * y), fma(B, f32(x * y), fma(A, f32(x * x), fma(q2, y, fma(q1, x, q0)))))
        alp = min(np.float64(f32(np.exp2(np.float64(pw)))), 0.999)
        S = 0.5 * (abs(a_) * dx * dx + abs(c_) * dy * dy) + abs(b_ * dx * dy)
        mxd, myd = np.float64(mx), np.float64(my)
        Sc = (0.5 * (abs(a_) * mxd * mxd + abs(c_) * myd * myd) + abs(b_ * mxd * myd) + abs((a_ * mxd + b_ * myd) * x) + abs((c_ * myd + b_ * mxd) * y)
              + 0.5 * (abs(a_) * x * x + abs(c_) * y * y) + abs(b_ * x * y))
        dl = abs(np.log(alp / al))
        w = al * T_ / max(1 - al, 1e-3)
        rows.append((dl, S, Sc, w, gid, np.hypot(dx, dy)))
        if T_ * (1 - al) <= 1e-4:
            break
        T_ *= (1 - al)
    rows.sort(key=lambda r: -r[0] * r[3])
    print("   contributors by |d ln alpha| x weight:  |d ln alpha|   /S        /S_c      weight   distance px")
    for dl, S, Sc, w, gid, dist in rows[:6]:
        print(f"      g {gid:7d}: {dl:.2e}   {dl / max(S, 1e-30):.2e}  {dl / Sc:.2e}   {w:.2e}  {dist:7.1f}")
    tot = sum(r[0] * r[3] for r in rows)
    print(f"   sum |d ln alpha| w = {tot:.2e};  sum 1e-6 S w = {sum(1e-6 * r[1] * r[3] for r in rows):.2e};  sum 3e-7 S_c w = {sum(3e-7 * r[2] * r[3] for r in rows):.2e}")
    allr = np.array([(r[0] / max(r[1], 1e-30), r[0] / r[2]) for r in rows])
    print(f"   over all {len(rows)} contributors: max |d ln alpha| / S = {allr[:, 0].max():.2e}, / S_c = {allr[:, 1].max():.2e}")
    # the same pixel through the three walks of soak_pixel_cause.py: is the kernel's error reproduced, and by which walk?
    import importlib.util
    spec = importlib.util.spec_from_file_location("spc", os.path.join(ROOT, "scripts", "dbg", "soak_pixel_cause.py"))
    src = open(spec.origin).read().split("\nfor seed in")[0]
    ns = {"__file__": spec.origin}
    exec(compile(src, spec.origin, "exec"), ns)
    res = {m: ns["walk"](px, py, lst, M2, CO, OP, FE, m) for m in ("f64", "f32", "poly")}
    print(f"   HIP {got[py, px]} alpha {ga[py, px]:.7f}")
    for m, (C, A, log) in res.items():
        print(f"   {m:5s} {np.asarray(C, np.float64)} alpha {float(A):.7f}  ({sum(1 for l in log if l[4] == 'take')} blended)")
