"""GPU parity of the backward path against autograd of the torch oracle
(oracle/gs_oracle_torch.py, float64 on the CPU), stage by stage and end to end.

Gradient tolerance: per-tensor, errors are scaled by the per-row magnitude plus 1e-3 of the
tensor's largest entry; the scaled error must stay below 2e-3 on all rows but a small
fraction (alpha-threshold flips move a whole Gaussian's contribution at one pixel), and the
cosine similarity of the full gradient must exceed 0.9999.
"""
import math

import numpy as np
import pytest
import torch

from oracle import gs_oracle_torch as OT
from robosimgs_amd import camera_ring, synthetic_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    return t.requires_grad_(grad)


def _d(a, grad=False):
    return torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=grad)


from grad_gate import compare as _compare, chained_budget      # noqa: E402  (tests/grad_gate.py: the gate and its rules)


def _scene(n, mu, deg, w, h, theta=0.3, seed=0):
    g = synthetic_scene(n, math.log(mu), deg, seed)
    cam = camera_ring(1, w, h, thetas=[theta])[0]
    return g, cam


def test_projection_backward():
    from robosimgs_amd import ops
    g, cam = _scene(4000, 0.1, 0, 160, 120)
    rng = np.random.default_rng(0)
    n = len(g)
    means, quats, scales = _t(g.means, True), _t(g.quats, True), _t(g.scales, True)
    vm = _t(cam.viewmat()[None], True)
    radii, m2d, dep, con, comp = ops.fully_fused_projection(means, None, quats, scales, vm,
                                                            _t(cam.K[None]), 160, 120,
                                                            calc_compensations=True)
    w1, w2, w3, w4 = (rng.normal(size=(n, 2)), rng.normal(size=n), rng.normal(size=(n, 3)),
                      rng.normal(size=n))
    vis = (radii[0] > 0)
    loss = ((m2d[0] * _t(w1)).sum(-1) + dep[0] * _t(w2) + (con[0] * _t(w3)).sum(-1)
            + comp[0] * _t(w4))[vis].sum()
    loss.backward()
    rm, rq, rs, rv = _d(g.means, True), _d(g.quats, True), _d(g.scales, True), _d(cam.viewmat(), True)
    p = OT.project(rm, rq, rs, rv, _d(cam.K), 160, 120)
    mask = torch.tensor(vis.cpu().numpy().astype(np.float64))
    rl = (((p["means2d"] * _d(w1)).sum(-1) + p["depths"] * _d(w2) + (p["conics"] * _d(w3)).sum(-1)
           + p["compensations"] * _d(w4)) * mask).sum()
    rl.backward()
    _compare("v_means", means.grad, rm.grad.numpy())
    _compare("v_quats", quats.grad, rq.grad.numpy())
    _compare("v_scales", scales.grad, rs.grad.numpy())
    _compare("v_viewmat", vm.grad[0, :3].reshape(1, 12), rv.grad.numpy()[:3].reshape(1, 12),
             row_tol=5e-3)


@pytest.mark.parametrize("deg,K", [(0, 1), (1, 4), (2, 9), (3, 16), (1, 16)])
def test_sh_backward(deg, K):
    from robosimgs_amd import ops
    rng = np.random.default_rng(deg)
    n = 1000
    dirs, coeffs, v = rng.normal(size=(n, 3)), rng.normal(size=(n, K, 3)), rng.normal(size=(n, 3))
    masks = rng.random(n) > 0.25
    d, c = _t(dirs, True), _t(coeffs, True)
    out = ops.spherical_harmonics(deg, d, c, torch.from_numpy(masks).to(DEV))
    (out * _t(v)).sum().backward()
    rd, rc = _d(dirs, True), _d(coeffs, True)
    ro = OT.spherical_harmonics(deg, rd, rc) * _d(masks.astype(np.float64))[:, None]
    (ro * _d(v)).sum().backward()
    _compare("v_coeffs", c.grad, rc.grad.numpy())
    if deg > 0:
        _compare("v_dirs", d.grad, rd.grad.numpy())
    else:
        assert float(d.grad.abs().max()) == 0.0


@pytest.mark.parametrize("n,mu,w,h,ch,bg", [(3000, 0.15, 96, 80, 3, False), (8000, 0.06, 128, 112, 4, True),
                                             (1500, 0.3, 70, 50, 1, False), (2000, 0.2, 64, 64, 7, True)])
def test_rasterize_backward(n, mu, w, h, ch, bg):
    """Blend-stage gradients on identical inputs (tile lists from the HIP binning)."""
    from robosimgs_amd import ops
    g, cam = _scene(n, mu, 0, w, h)
    rng = np.random.default_rng(1)
    radii, m2d, dep, con, _ = ops.fully_fused_projection(
        _t(g.means), None, _t(g.quats), _t(g.scales), _t(cam.viewmat()[None]), _t(cam.K[None]), w, h)
    tw, th = -(-w // 16), -(-h // 16)
    _, isect_ids, flat = ops.isect_tiles(m2d, radii, dep, 16, tw, th)
    offs = ops.isect_offset_encode(isect_ids, 1, tw, th)
    colors_np = rng.random((n, ch))
    opac_np = g.opacities
    bg_np = rng.random(ch) if bg else None
    wr, wa = rng.normal(size=(h, w, ch)), rng.normal(size=(h, w))
    a_m2d, a_con = m2d.detach().clone().requires_grad_(True), con.detach().clone().requires_grad_(True)
    a_col, a_op = _t(colors_np[None], True), _t(opac_np[None], True)
    a_bg = _t(bg_np[None], True) if bg else None
    render, alphas = ops.rasterize_to_pixels(a_m2d, a_con, a_col, a_op, w, h, 16, offs, flat,
                                             backgrounds=a_bg)
    ((render[0] * _t(wr)).sum() + (alphas[0, ..., 0] * _t(wa)).sum()).backward()
    r_m2d, r_con = _d(m2d[0].cpu().numpy(), True), _d(con[0].cpu().numpy(), True)
    r_col, r_op = _d(colors_np, True), _d(opac_np, True)
    r_bg = _d(bg_np, True) if bg else None
    img, al = OT.rasterize(r_m2d, r_con, r_col, r_op, flat.cpu().numpy(),
                           offs[0].cpu().numpy(), w, h, 16, r_bg)
    np.testing.assert_allclose(render[0].detach().cpu().numpy(), img.detach().numpy(), atol=2e-4)
    ((img * _d(wr)).sum() + (al * _d(wa)).sum()).backward()
    _compare("v_means2d", a_m2d.grad[0], r_m2d.grad.numpy(), bad_frac=5e-3)
    _compare("v_conics", a_con.grad[0], r_con.grad.numpy(), bad_frac=5e-3)
    _compare("v_colors", a_col.grad[0], r_col.grad.numpy(), bad_frac=5e-3)
    _compare("v_opacities", a_op.grad[0], r_op.grad.numpy().reshape(-1, 1), bad_frac=5e-3)
    if bg:
        _compare("v_backgrounds", a_bg.grad, r_bg.grad.numpy().reshape(1, -1))


def test_rasterize_absgrad():
    from robosimgs_amd import ops
    g, cam = _scene(2000, 0.15, 0, 64, 64)
    radii, m2d, dep, con, _ = ops.fully_fused_projection(
        _t(g.means), None, _t(g.quats), _t(g.scales), _t(cam.viewmat()[None]), _t(cam.K[None]), 64, 64)
    _, isect_ids, flat = ops.isect_tiles(m2d, radii, dep, 16, 4, 4)
    offs = ops.isect_offset_encode(isect_ids, 1, 4, 4)
    a_m2d = m2d.detach().clone().requires_grad_(True)
    col = torch.rand(1, len(g), 3, device=DEV)
    render, _ = ops.rasterize_to_pixels(a_m2d, con, col, _t(g.opacities[None]), 64, 64, 16, offs,
                                        flat, absgrad=True)
    render.sum().backward()
    assert hasattr(a_m2d, "absgrad")
    # |sum| <= sum |.| up to rounding (signed sums: tile-centred moments; absolute ones: per pixel)
    assert torch.all(a_m2d.absgrad * (1 + 2e-5) + 2e-6 >= a_m2d.grad.abs()), float((a_m2d.grad.abs() - a_m2d.absgrad).max())


@pytest.mark.parametrize("deg,mode,aa,seg", [(0, "RGB", False, 256), (3, "RGB", False, 64), (2, "RGB+ED", False, 64),
                                             (1, "RGB", True, 64), (3, "RGB+D", True, 0), (2, "RGB+ED", False, 0)])
def test_rasterization_backward_end_to_end(deg, mode, aa, seg):
    """Whole path (BASELINE configs[2] shape at test size): L = <w, render> + <u, alpha>.  seg = backward_segment:
    64 cuts this scene's lists (up to ~350 entries) into up to six segments, 0 is the whole-list walk."""
    from robosimgs_amd import rasterization
    w, h = 112, 80
    g, cam = _scene(6000, 0.07, deg, w, h)
    t = g.to_torch(DEV, deg)
    names = ["means", "quats", "scales", "opacities", "colors"]
    for k in names:
        t[k].requires_grad_(True)
    rm = "antialiased" if aa else "classic"
    colors, alphas, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"],
                                         t["colors"], _t(cam.viewmat()[None]), _t(cam.K[None]),
                                         w, h, sh_degree=deg, render_mode=mode, rasterize_mode=rm, backward_segment=seg)
    rng = np.random.default_rng(2)
    wr, wa = rng.normal(size=tuple(colors.shape[1:])), rng.normal(size=(h, w))
    ((colors[0] * _t(wr)).sum() + (alphas[0, ..., 0] * _t(wa)).sum()).backward()
    r = {k: _d(v, True) for k, v in (("means", g.means), ("quats", g.quats), ("scales", g.scales),
                                      ("opacities", g.opacities),
                                      ("colors", g.sh_coeffs[:, :(deg + 1) ** 2]))}
    img, al, _ = OT.render(r["means"], r["quats"], r["scales"], r["opacities"], r["colors"],
                           _d(cam.viewmat()), _d(cam.K), w, h, sh_degree=deg, render_mode=mode,
                           rasterize_mode=rm)
    ((img * _d(wr)).sum() + (al[..., 0] * _d(wa)).sum()).backward()
    # EVERY row must lie within rounding + 1.5 x its flip budget of the oracle's: what the near-flip decisions of the
    # fp64 blend are worth at the could-flip pixels the Gaussian reaches (oracle/gs_cpu.cpp Extras::budget, chained
    # through projection and SH with the absolute Jacobian; tests/grad_gate.py).  The port has no anti-aliased mode:
    # those cases keep the fraction bound only.
    budgets = {k: None for k in names}
    if not aa:
        from grad_gate import oracle_budgets, parameter_budgets
        from oracle import gs_oracle_np as O
        f32 = lambda m: np.asarray(m, dtype=np.float32)
        info = oracle_budgets(g, f32(cam.viewmat()), f32(cam.K), w, h, deg, mode, wr, wa, O.EPS_PATH_GRAD)
        bud = info["budget"]
        # the blend stage on its own, all four of its outputs
        bg = meta["blend_grads"][0]
        for name, got, ref, b in (("means2d", bg[0], info["g_means2d"], bud[:, 0]), ("conics", bg[1], info["g_conics"], bud[:, 1]),
                                  ("feats", bg[2], info["g_feats"], bud[:, 2]),
                                  ("opacities", bg[3], info["g_opacities"].reshape(-1, 1), bud[:, 3])):
            _compare("blend v_" + name, got, ref, row_tol=5e-3, bad_frac=1e-2, cos_min=0.999, budget=b)
        budgets.update(parameter_budgets(g, f32(cam.viewmat()), f32(cam.K), w, h, deg, mode != "RGB", bud))
        budgets["opacities"] = bud[:, 3]
    for k in names:
        _compare("v_" + k, t[k].grad, r[k].grad.numpy() if r[k].grad.ndim > 1
                 else r[k].grad.numpy().reshape(-1, 1), row_tol=5e-3, bad_frac=1e-2, cos_min=0.999, budget=budgets[k])


def test_multi_camera_gradients_accumulate():
    from robosimgs_amd import rasterization
    g = synthetic_scene(3000, math.log(0.1), 1, 5)
    cams = camera_ring(2, 96, 64)
    vm = _t(np.stack([c.viewmat() for c in cams]))
    Ks = _t(np.stack([c.K for c in cams]))

    def grads(sel):
        t = g.to_torch(DEV, 1)
        for k in ("means", "quats", "scales", "opacities", "colors"):
            t[k].requires_grad_(True)
        out, al, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                                   vm[sel], Ks[sel], 96, 64, sh_degree=1)
        (out.sum() + al.sum()).backward()
        return [t[k].grad.clone() for k in ("means", "quats", "scales", "opacities", "colors")]
    both, a, b = grads(slice(0, 2)), grads(slice(0, 1)), grads(slice(1, 2))
    for x, y, z in zip(both, a, b):
        torch.testing.assert_close(x, y + z, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("mode,aa,bg,seg", [("RGB+ED", False, True, 64), ("RGB", True, False, 256), ("RGB+D", False, False, 0)])
def test_batched_training_cameras_through_one_call_equal_the_per_camera_loop(mode, aa, bg, seg):
    """rasterization() with gradients and a fixed list capacity sends the whole camera batch through
    mgs_render_frames_train / mgs_render_frames_backward (two C calls instead of five per camera): frames, every
    gradient, the per-camera screen-space gradients and the meta arrays are those of the per-camera entry points, bit
    for bit; camera-pose gradients agree to the order of their atomics."""
    from robosimgs_amd import rasterization
    g = synthetic_scene(9000, math.log(0.09), 2, 5)
    cams = camera_ring(3, 144, 96)
    vm0 = _t(np.stack([c.viewmat() for c in cams]))
    Ks = _t(np.stack([c.K for c in cams]))
    ch = 3 if mode == "RGB" else 4
    gen = torch.Generator(DEV).manual_seed(8)
    w_c = torch.randn(3, 96, 144, ch, device=DEV, generator=gen)
    w_a = torch.randn(3, 96, 144, 1, device=DEV, generator=gen)
    bgs = torch.rand(3, ch, device=DEV, generator=gen) if bg else None
    names = ("means", "quats", "scales", "opacities", "colors")

    def run(cap):
        t = g.to_torch(DEV, 2)
        p = {k: t[k].detach().clone().requires_grad_(True) for k in names}
        vm = vm0.clone().requires_grad_(True)
        c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, Ks, 144, 96,
                                   sh_degree=2, render_mode=mode, rasterize_mode="antialiased" if aa else "classic",
                                   backgrounds=bgs, absgrad=True, isect_capacity=cap, backward_segment=seg)
        meta["means2d"].retain_grad()
        ((c * w_c).sum() + (a * w_a).sum()).backward()
        return c.detach(), a.detach(), [p[k].grad for k in names], vm.grad, meta

    c0, a0, g0, v0, m0 = run(None)               # per-camera entry points (the capacity is read back per camera)
    c1, a1, g1, v1, m1 = run(600_000)            # the batch behind two C calls
    assert torch.equal(c0, c1) and torch.equal(a0, a1)
    for k, x, y in zip(names, g0, g1):
        assert torch.equal(x, y), k
    torch.testing.assert_close(v0, v1, rtol=1e-4, atol=1e-5)
    assert torch.equal(m0["means2d"].grad, m1["means2d"].grad) and torch.equal(m0["means2d"].absgrad, m1["means2d"].absgrad)
    for key in ("radii", "means2d", "depths", "conics", "tiles_per_gauss", "n_isects", "isect_offsets"):
        assert torch.equal(m0[key], m1[key]), key
    n_is = [int(x) for x in m1["n_isects"]]
    assert all(torch.equal(m0["tile_lists"][c].flatten_ids[:n_is[c]], m1["tile_lists"][c].flatten_ids[:n_is[c]]) for c in range(3))
    assert int(m1["isect_status"].max()) == 0 and len(m1["flatten_ids"]) == sum(n_is)     # gsplat's flat lists still assemble


def test_deterministic_backward_matches_atomic_and_is_bit_reproducible():
    """mgs_rasterize_bwd_det (records + per-Gaussian reduce) vs mgs_rasterize_bwd (atomics):
    same sums up to float re-association; two det runs are bit-identical."""
    from robosimgs_amd import ops
    g, cam = _scene(9000, 0.08, 0, 144, 96)
    t = g.to_torch(DEV, 0)
    vm, K = _t(cam.viewmat()), _t(cam.K)
    radii, m2d, dep, con, _, feats = ops.project_color_fwd_raw(
        t["means"], t["quats"], t["scales"], t["opacities"], 0, t["colors"], vm, K, 144, 96, 0.3,
        0.01, 1e10, 0.0, False, True)
    tw, th = 9, 6
    tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 400_000, want_pair_info=True)
    bg = torch.tensor([0.3, 0.1, 0.2, 0.0], device=DEV)
    out = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], bg, 144, 96, tw, th, tl.tile_offsets,
                                tl.flatten_ids)
    vr, va = torch.randn(96, 144, 4, device=DEV), torch.randn(96, 144, device=DEV)
    a = ops.rasterize_bwd_raw(m2d, con, feats, t["opacities"], bg, 144, 96, tw, th, tl.tile_offsets,
                              tl.flatten_ids, out[1], out[2], vr, va, absgrad=True)
    d1 = ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], bg, 144, 96, tw, th, tl, out[1],
                                   out[2], vr, va, absgrad=True)
    d2 = ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], bg, 144, 96, tw, th, tl, out[1],
                                   out[2], vr, va, absgrad=True)
    for x, y, z in zip(a, d1, d2):
        assert torch.equal(y, z)                                     # bit-reproducible
        scale = float(x.abs().max()) + 1e-20
        assert float((x - y).abs().max()) / scale < 1e-4
    # the slot map is a bijection onto [0, n_isect)
    info = tl.pair_info.cpu().numpy().astype(np.int64)
    cnt = (info[:, 3] & 0xffff) * (info[:, 3] >> 16)
    assert cnt.sum() == int(tl.n_isect.item())
    vis = cnt > 0
    order = np.argsort(info[vis, 0])
    starts, sizes = info[vis, 0][order], cnt[vis][order]
    assert starts[0] == 0 and np.all(starts[1:] == starts[:-1] + sizes[:-1])


def test_screen_space_gradients_are_published():
    """meta["means2d_grad"] / ["means2d_absgrad"] after backward (densification inputs)."""
    from robosimgs_amd import rasterization
    g, cam = _scene(3000, 0.1, 1, 96, 64)
    t = g.to_torch(DEV, 1)
    t["means"].requires_grad_(True)
    c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                               _t(cam.viewmat()[None]), _t(cam.K[None]), 96, 64, sh_degree=1, absgrad=True)
    assert "means2d_grad" not in meta
    meta["means2d"].retain_grad()                      # what splatfacto does; must not raise
    c.sum().backward()
    g2d, gabs = meta["means2d_grad"][0], meta["means2d_absgrad"][0]
    assert torch.equal(meta["means2d"].grad[0], g2d) and torch.equal(meta["means2d"].absgrad[0], gabs)
    assert g2d.shape == (3000, 2) and float(g2d.abs().sum()) > 0
    # |sum| <= sum |.|, up to rounding: the signed sums go through the tile-centred moments (raster_bwd.hip
    # moments_to_mean), the absolute ones are summed per pixel
    assert bool((gabs * (1 + 2e-5) + 2e-6 >= g2d.abs()).all()), float((g2d.abs() - gabs).max())
    assert bool((g2d[meta["radii"][0] == 0] == 0).all())


@pytest.mark.parametrize("case", ["elongated", "faint", "huge", "config1", "antialiased"])
def test_tight_tile_bounds_change_no_bit(case):
    """mgs_isect_tiles with conics + opacities drops (tile, Gaussian) pairs that cannot reach
    alpha >= 1/255 at any pixel centre.  Image, alpha and every gradient must be bit-identical to
    the classic rectangles, for needle-like footprints (fp32 cancellation in the conic), barely
    visible opacities, screen-filling Gaussians and the anti-aliased opacity."""
    from robosimgs_amd import rasterization
    W, H = 200, 136
    rng = np.random.default_rng(7)
    g = synthetic_scene(6000, math.log(0.05), 1, 11)
    kw = {}
    if case == "elongated":
        g.log_scales[:, 0] += 2.5
        g.log_scales[:, 1:] -= 2.0
    elif case == "faint":
        g.opacity_logits[:] = rng.normal(-5.3, 0.6, size=g.opacity_logits.shape)   # around 1/255
    elif case == "huge":
        g.log_scales[::50] += 3.5
    elif case == "antialiased":
        g.log_scales[:] -= 1.5
        kw["rasterize_mode"] = "antialiased"
    cam = camera_ring(1, W, H, thetas=[0.7])[0]
    t = g.to_torch(DEV, 1)
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    w_img = torch.from_numpy(rng.normal(size=(H, W, 4)).astype(np.float32)).to(DEV)
    w_a = torch.from_numpy(rng.normal(size=(H, W, 1)).astype(np.float32)).to(DEV)
    outs = {}
    for bounds in ("classic", "tight"):
        p = {k: t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
        # (backward_segment=0: the whole-list walk.  Segments start at multiples of 128 LISTED entries, which the two
        #  list flavours place differently, so segmented gradients agree to rounding only -- checked below)
        c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K,
                                   W, H, sh_degree=1, render_mode="RGB+D", absgrad=True,
                                   tile_bounds=bounds, backward_segment=0, **kw)
        ((c[0] * w_img).sum() + (a[0] * w_a).sum()).backward()
        outs[bounds] = (c.detach(), a.detach(), {k: v.grad for k, v in p.items()},
                        meta["means2d"].absgrad, int(meta["n_isects"][0]), meta["tiles_per_gauss"])
    cc, ca, cg, cabs, cn, ctp = outs["classic"]
    tc, ta, tg, tabs, tn, ttp = outs["tight"]
    assert torch.equal(cc, tc) and torch.equal(ca, ta)
    for k in cg:
        assert torch.equal(cg[k], tg[k]), k
    assert torch.equal(cabs, tabs)
    assert tn < cn and bool((ttp <= ctp).all()), (tn, cn)
    # the segmented walk (the default) on either list flavour: the same gradients up to rounding
    for bounds in ("classic", "tight"):
        p = {k: t[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
        c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K,
                                   W, H, sh_degree=1, render_mode="RGB+D", absgrad=True, tile_bounds=bounds,
                                   backward_segment=64, **kw)
        ((c[0] * w_img).sum() + (a[0] * w_a).sum()).backward()
        assert torch.equal(c.detach(), cc) and torch.equal(a.detach(), ca)
        for k in cg:
            # (against fp64 autograd the segmented walk is the CLOSER of the two in every case here -- it restarts from
            #  the forward's exact T where the whole-list walk keeps multiplying v_rcp_f32 results: scripts/dbg/
            #  split_elongated.py, needles: quats 2.8e-3 against 4.5e-3 of the largest entry; they differ by 2.2e-3)
            scale = float(cg[k].abs().max())
            assert float((p[k].grad - cg[k]).abs().max()) <= 5e-3 * scale, (bounds, k)


@pytest.mark.parametrize("mode,bg,seg", [("RGB+ED", True, 64), ("RGB", False, 64), ("RGB+D", True, 128), ("RGB+ED", False, 256)])
def test_segmented_backward_matches_whole_list_walk(mode, bg, seg):
    """The backward walks a tile's list as independent segments that start from the forward's checkpoints
    (include/mgs.h: mgs_rasterize_fwd checkpoints).  Dense scene: lists of several hundred entries, so tiles have
    3-10 segments.  Image and alpha are untouched by the checkpoint stores; gradients equal the whole-list walk's up
    to rounding (the forward's T instead of a chain of reciprocals) and are bit-reproducible."""
    from robosimgs_amd import rasterization
    W, H = 176, 120
    g = synthetic_scene(20000, math.log(0.06), 2, 3)
    g.opacity_logits[:] -= 2.0                      # faint: pixels stay open deep into the lists
    cam = camera_ring(1, W, H, thetas=[1.1])[0]
    t = g.to_torch(DEV, 2)
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    names = ("means", "quats", "scales", "opacities", "colors")
    ch = 3 if mode == "RGB" else 4
    gen = torch.Generator(DEV).manual_seed(5)
    w_c = torch.randn(1, H, W, ch, device=DEV, generator=gen)
    w_a = torch.randn(1, H, W, 1, device=DEV, generator=gen)
    bgs = torch.rand(1, ch, device=DEV, generator=gen) if bg else None

    def run(segment):
        p = {k: t[k].detach().clone().requires_grad_(True) for k in names}
        c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, W, H,
                                   sh_degree=2, render_mode=mode, backgrounds=bgs, backward_segment=segment)
        ((c * w_c).sum() + (a * w_a).sum()).backward()
        return c.detach(), a.detach(), [p[k].grad for k in names], meta

    c0, a0, g0, meta = run(0)
    lens = (meta["tile_lists"][0].tile_offsets[1:] - meta["tile_lists"][0].tile_offsets[:-1])
    assert int(lens.max()) > 3 * seg, int(lens.max())          # the case really has tiles of several segments
    c1, a1, g1, _ = run(seg)
    c2, a2, g2, _ = run(seg)
    assert torch.equal(c0, c1) and torch.equal(a0, a1)
    for k, x, y, z in zip(names, g0, g1, g2):
        assert torch.equal(y, z), k                              # bit-reproducible
        assert torch.isfinite(y).all()
        scale = float(x.abs().max())
        assert float((x - y).abs().max()) <= 1e-3 * scale, (k, float((x - y).abs().max()), scale)
    assert any(not torch.equal(x, y) for x, y in zip(g0, g1))   # ... and it really is another walk


def test_opacity_zero_under_classic_bounds_gives_finite_gradients():
    """A Gaussian with opacity exactly 0 owns record slots under classic tile bounds; its opacity gradient is 0, not
    0 / 0 (the reduce divides sum v_sigma by the opacity once per Gaussian)."""
    from robosimgs_amd import rasterization
    g, cam = _scene(3000, 0.12, 1, 96, 64)
    t = g.to_torch(DEV, 1)
    with torch.no_grad():
        t["opacities"][::7] = 0.0
    p = {k: t[k].detach().clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")}
    c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"],
                               _t(cam.viewmat())[None], _t(cam.K)[None], 96, 64, sh_degree=1, tile_bounds="classic")
    assert int(meta["tiles_per_gauss"][0][::7].sum()) > 0        # they do sit in tile rectangles
    (c.sum() + a.sum()).backward()
    for k, v in p.items():
        assert torch.isfinite(v.grad).all(), k
    assert float(p["opacities"].grad[::7].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(1, 37, 53, 3), (1080, 1920, 3), (7,), (4, 4)])
def test_fused_l1_loss_matches_torch(shape):
    from robosimgs_amd import l1_loss
    gen = torch.Generator(DEV).manual_seed(3)
    a = torch.rand(*shape, device=DEV, generator=gen).requires_grad_(True)
    b = torch.rand(*shape, device=DEV, generator=gen)
    with torch.no_grad():
        b.view(-1)[::5] = a.view(-1)[::5]                      # exact zeros: sign(0) = 0
    loss = l1_loss(a, b)
    (loss * 3.0).backward()
    a2 = a.detach().clone().requires_grad_(True)
    ref = (a2 - b).abs().mean()
    (ref * 3.0).backward()
    assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, ref.item())
    torch.testing.assert_close(a.grad, a2.grad, rtol=1e-6, atol=0)
    assert l1_loss(a, b).item() == loss.item()                  # fixed summation order
    assert l1_loss(a.detach(), b).item() == loss.item()         # the forward-only kernels: the same bits
    # grad_output 1 (the scale launch returns at once), and a second backward through a retained graph
    a3 = a.detach().clone().requires_grad_(True)
    l3 = l1_loss(a3, b)
    l3.backward(retain_graph=True)
    torch.testing.assert_close(a3.grad * 3.0, a2.grad, rtol=1e-6, atol=0)
    l3.backward()
    torch.testing.assert_close(a3.grad * 1.5, a2.grad, rtol=1e-6, atol=0)
    # unit_gradient: the cached 1.0 as the cotangent (no fill launch, no scale launch): the gradient of a plain backward()
    from robosimgs_amd import unit_gradient
    a4 = a.detach().clone().requires_grad_(True)
    l4 = l1_loss(a4, b)
    l4.backward(gradient=unit_gradient(l4))
    assert torch.equal(a4.grad * 3.0, a.grad) or torch.allclose(a4.grad * 3.0, a.grad, rtol=1e-6, atol=0)
    assert float(unit_gradient(l4)) == 1.0 and unit_gradient(l4) is unit_gradient(l4)
    # a contiguous view at an odd element offset
    if len(shape) == 1:
        v = torch.rand(9, device=DEV)[1:8]
        assert abs(float(l1_loss(v, b)) - float((v - b).abs().mean())) < 1e-6


@pytest.mark.parametrize("deg,mode", [(0, "RGB"), (3, "RGB+ED")])
def test_camera_pose_gradient_matches_autograd(deg, mode):
    """d loss / d viewmat through the fused path (projection AND the SH view direction), against
    autograd of the fp64 torch oracle; two cameras, each with its own 4x4 gradient."""
    from robosimgs_amd import rasterization
    w, h = 96, 64
    g = synthetic_scene(3000, math.log(0.09), deg, 4)
    cams = camera_ring(2, w, h, thetas=[0.3, 1.9])
    t = g.to_torch(DEV, deg)
    vm = _t(np.stack([c.viewmat() for c in cams]), True)
    Ks = _t(np.stack([c.K for c in cams]))
    colors, alphas, _ = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                                      vm, Ks, w, h, sh_degree=deg, render_mode=mode)
    rng = np.random.default_rng(6)
    wr, wa = rng.normal(size=tuple(colors.shape)), rng.normal(size=(2, h, w))
    ((colors * _t(wr)).sum() + (alphas[..., 0] * _t(wa)).sum()).backward()
    assert vm.grad is not None and vm.grad.shape == (2, 4, 4)
    assert float(vm.grad[:, 3].abs().max()) == 0.0                      # the [0 0 0 1] row has no gradient
    for c, cam in enumerate(cams):
        rv = _d(cam.viewmat(), True)
        img, al, _ = OT.render(_d(g.means), _d(g.quats), _d(g.scales), _d(g.opacities),
                               _d(g.sh_coeffs[:, :(deg + 1) ** 2]), rv, _d(cam.K), w, h, sh_degree=deg,
                               render_mode=mode)
        ((img * _d(wr[c])).sum() + (al[..., 0] * _d(wa[c])).sum()).backward()
        ref = rv.grad.numpy()[:3]
        got = vm.grad[c, :3].cpu().double().numpy()
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max(), (c, got, ref)


@pytest.mark.parametrize("ch", [5, 12, 20, 32])
def test_deterministic_backward_many_channels(ch):
    """Record-based backward with more than 4 feature channels (several 8-value LDS reduction
    rounds plus a DPP remainder per record) against the atomic backward."""
    from robosimgs_amd import ops
    g, cam = _scene(4000, 0.1, 0, 112, 80)
    t = g.to_torch(DEV, 0)
    vm, K = _t(cam.viewmat()), _t(cam.K)
    radii, m2d, dep, con, _ = ops.projection_fwd_raw(t["means"], t["quats"], t["scales"], vm, K, 112, 80,
                                                     0.3, 0.01, 1e10, 0.0, False)
    tw, th = 7, 5
    tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, 300_000, want_pair_info=True)
    gen = torch.Generator(DEV).manual_seed(ch)
    feats = torch.rand(len(g), ch, device=DEV, generator=gen)
    bg = torch.rand(ch, device=DEV, generator=gen)
    out = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], bg, 112, 80, tw, th, tl.tile_offsets,
                                tl.flatten_ids)
    # the inference variant (no last_ids, four tiles per workgroup) renders the same bits
    inf = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], bg, 112, 80, tw, th, tl.tile_offsets,
                                tl.flatten_ids, track_last=False)
    assert inf[2] is None and torch.equal(inf[0], out[0]) and torch.equal(inf[1], out[1])
    vr = torch.randn(80, 112, ch, device=DEV, generator=gen)
    va = torch.randn(80, 112, device=DEV, generator=gen)
    a = ops.rasterize_bwd_raw(m2d, con, feats, t["opacities"], bg, 112, 80, tw, th, tl.tile_offsets,
                              tl.flatten_ids, out[1], out[2], vr, va, absgrad=True)
    d = ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], bg, 112, 80, tw, th, tl, out[1], out[2],
                                  vr, va, absgrad=True)
    for x, y in zip(a, d):
        scale = float(x.abs().max()) + 1e-20
        assert float((x - y).abs().max()) / scale < 1e-4


def test_deterministic_backward_with_overflowed_lists_stays_inside_its_workspace():
    """include/mgs.h: on a capacity overflow "nothing is written out of bounds".  The record
    backward's slot bases run up to the TRUE n_isect; with an undersized capacity the kernels must
    skip every slot at or past it.  Canary bytes behind the workspace must survive, and the status
    word must say that the gradients are not to be trusted."""
    from robosimgs_amd import ops
    g, cam = _scene(6000, 0.12, 1, 160, 128)
    t = g.to_torch(DEV, 1)
    vm, K = _t(cam.viewmat()), _t(cam.K)
    radii, m2d, dep, con, _, feats, splats = ops.project_color_fwd_raw(
        t["means"], t["quats"], t["scales"], t["opacities"], 1, t["colors"], vm, K, 160, 128, 0.3, 0.01,
        1e10, 0.0, False, False, want_splats=True)
    tw, th = 10, 8
    need = int(ops.isect_tiles_raw(m2d, radii, dep, tw, th, 1_000_000).n_isect.item())
    for cap in (need // 3, 1000, 7):
        tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, cap, want_pair_info=True)
        assert int(tl.status.item()) != 0 and int(tl.n_isect.item()) == need
        render, alphas, last = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, 160, 128, tw, th,
                                                     tl.tile_offsets, tl.flatten_ids, splats=splats)
        v_r, v_a = torch.randn_like(render), torch.randn_like(alphas)
        out = ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, 160, 128, tw, th, tl, alphas,
                                        last, v_r, v_a, splats=splats, canary_bytes=1 << 20)
        torch.cuda.synchronize()
        assert bool((out[5] == 0xA5).all()), f"capacity {cap}: the backward wrote past its workspace"
        # the segmented walk on the same cut lists: checkpoints and unit tables are sized by the capacity too
        n_ck = ops.checkpoint_buffer(cap, tw, th, 3, 64, DEV).numel()
        ck = torch.full((n_ck + 65536,), 1234.5, device=DEV)
        render, alphas, last = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, 160, 128, tw, th,
                                                     tl.tile_offsets, tl.flatten_ids, splats=splats, latency=True,
                                                     checkpoints=ck[:n_ck], checkpoint_interval=64)
        out = ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, 160, 128, tw, th, tl, alphas,
                                        last, v_r, v_a, splats=splats, canary_bytes=1 << 20, render_out=render,
                                        checkpoints=ck[:n_ck], checkpoint_interval=64)
        torch.cuda.synchronize()
        assert bool((ck[n_ck:] == 1234.5).all()), f"capacity {cap}: the forward wrote past its checkpoint buffer"
        assert bool((out[5] == 0xA5).all()), f"capacity {cap}: the segmented backward wrote past its workspace"
        assert all(bool(torch.isfinite(x).all()) for x in out[:4])


@pytest.mark.parametrize("mode", ["RGB", "RGB+ED"])
def test_unused_outputs_need_no_zero_cotangent(mode):
    """A loss on the colours alone (or on alpha alone) reaches the backward with None for the other output
    (set_materialize_grads(False)): no zero frame is allocated, filled and read.  Gradients must equal bit for bit
    those of the same loss with the unused output added at weight zero."""
    from robosimgs_amd import rasterization
    g, cam = _scene(3000, 0.15, 2, 96, 64)
    t = g.to_torch(DEV, 2)
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    names = ("means", "quats", "scales", "opacities", "colors")
    w_c = torch.rand(1, 64, 96, 4 if "D" in mode else 3, device=DEV)
    w_a = torch.rand(1, 64, 96, 1, device=DEV)

    def grads(loss_of):
        p = {k: t[k].detach().clone().requires_grad_(True) for k in names}
        c, a, _ = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, 96, 64,
                                sh_degree=2, render_mode=mode)
        loss_of(c, a).backward()
        return [p[k].grad for k in names]

    for only, both in ((lambda c, a: (c * w_c).sum(), lambda c, a: (c * w_c).sum() + (a * 0.0).sum()),
                       (lambda c, a: (a * w_a).sum(), lambda c, a: (a * w_a).sum() + (c * 0.0).sum())):
        for x, y in zip(grads(only), grads(both)):
            assert torch.equal(x, y)


def test_reordering_a_trainers_parameters_changes_nothing_but_memory_order():
    """robosimgs_amd.reorder_parameters: parameters AND Adam's moments go into Morton order of the means, in place (the
    optimiser keeps its tensors); the next optimiser steps are those of the un-reordered run, row for row."""
    from robosimgs_amd import l1_loss, rasterization, reorder_parameters
    g = synthetic_scene(8000, math.log(0.08), 2, 12)
    cam = camera_ring(1, 160, 112, thetas=[0.5])[0]
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    names = ("means", "quats", "scales", "opacities", "colors")
    target = torch.rand(1, 112, 160, 3, device=DEV, generator=torch.Generator(DEV).manual_seed(2))

    def make():
        t = g.to_torch(DEV, 2)
        p = {k: t[k].detach().clone().requires_grad_(True) for k in names}
        return p, torch.optim.Adam(list(p.values()), lr=1e-3)

    def step(p, opt):
        opt.zero_grad(set_to_none=True)
        c, a, _ = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, 160, 112, sh_degree=2)
        l1_loss(c, target).backward()
        opt.step()

    pa, oa = make()
    pb, ob = make()
    for _ in range(2):
        step(pa, oa); step(pb, ob)
    ids = {k: pb[k].data_ptr() for k in names}
    order = reorder_parameters(pb, ob)
    assert sorted(order.tolist()) == list(range(8000)) and not torch.equal(order, torch.arange(8000, device=DEV))
    assert all(pb[k].data_ptr() == ids[k] for k in names)                     # in place
    assert torch.equal(pb["means"], pa["means"][order])
    for _ in range(2):
        step(pa, oa); step(pb, ob)
    for k in names:       # same trajectory up to the summation order of a Gaussian's pixels (depth ties, float re-association)
        torch.testing.assert_close(pb[k], pa[k][order], rtol=2e-4, atol=2e-6)


def test_trainer_helper_renders_in_morton_order_by_itself_on_the_same_trajectory():
    """robosimgs_amd.Trainer: the INTEGRATION.md loop with one object -- the parameters go into Morton order at the first render
    and every `auto_reorder_every` steps without the caller doing anything, the run follows the plain rasterization() loop
    (up to the summation order of a Gaussian's pixels), original_index maps back."""
    from robosimgs_amd import Trainer, l1_loss, rasterization
    g = synthetic_scene(8000, math.log(0.08), 2, 12)
    cam = camera_ring(1, 160, 112, thetas=[0.5])[0]
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    names = ("means", "quats", "scales", "opacities", "colors")
    target = torch.rand(1, 112, 160, 4, device=DEV, generator=torch.Generator(DEV).manual_seed(2))

    def make():
        t = g.to_torch(DEV, 2)
        p = {k: t[k].detach().clone().requires_grad_(True) for k in names}
        return p, torch.optim.Adam(list(p.values()), lr=1e-3)

    pa, oa = make()
    for _ in range(5):
        oa.zero_grad(set_to_none=True)
        c, a, _ = rasterization(pa["means"], pa["quats"], pa["scales"], pa["opacities"], pa["colors"], vm, K, 160, 112, sh_degree=2,
                                render_mode="RGB+ED", isect_capacity=400_000)
        l1_loss(c, target).backward()
        oa.step()
    pb, ob = make()
    tr = Trainer(pb, ob, 160, 112, auto_reorder_every=2, sh_degree=2, render_mode="RGB+ED", isect_capacity=400_000)
    losses = []
    for _ in range(5):
        c, a, meta = tr.render(vm, K)
        loss = l1_loss(c, target)
        losses.append(float(loss))
        tr.step(loss)
    assert tr.reorders == 3 and losses[-1] < losses[0]
    assert not torch.equal(tr.original_index, torch.arange(8000, device=DEV))
    for k in names:
        torch.testing.assert_close(tr.in_original_order(pb[k].detach()), pa[k].detach(), rtol=2e-4, atol=2e-6)


def test_splat_slot_words_of_another_binning_are_not_trusted():
    """mgs_isect_tiles(splat_slots=) writes a binning's record slots into the splat records in place; rasterize_bwd_det_raw takes
    them from there only while the records still belong to THAT binning (ops._splat_slots_valid): binned again under another
    rule -- other slots in the same words -- an older TileLists falls back to its own pair_info and gives the same gradients."""
    from robosimgs_amd import ops
    g, cam = _scene(6000, 0.1, 1, 160, 112)
    t = g.to_torch(DEV, 1)
    vm, K = _t(cam.viewmat()), _t(cam.K)
    W, H, tw, th = 160, 112, 10, 7
    radii, m2d, dep, con, _, feats, splats = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], 1, t["colors"], vm, K, W, H,
                                                                       0.3, 0.01, 1e10, 0.0, False, True, want_splats=True)
    cap = ops._upper_bound_isects(radii, tw, th) + 1
    tl_tight = ops.isect_tiles_raw(m2d, radii, dep, tw, th, cap, want_pair_info=True, conics=con, opacities=t["opacities"], splats=splats)
    assert ops._splat_slots_valid(tl_tight, splats)
    r, a, last = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl_tight.tile_offsets, tl_tight.flatten_ids, splats=splats)
    vr, va = torch.randn(H, W, 4, device=DEV), torch.randn(H, W, device=DEV)
    want = ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl_tight, a, last, vr, va, splats=splats)[:4]
    # the same records binned again with CLASSIC rectangles: their slot words now belong to that binning
    tl_classic = ops.isect_tiles_raw(m2d, radii, dep, tw, th, cap, want_pair_info=True, splats=splats)
    assert ops._splat_slots_valid(tl_classic, splats) and not ops._splat_slots_valid(tl_tight, splats)
    assert not torch.equal(tl_classic.pair_info, tl_tight.pair_info)
    got = ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl_tight, a, last, vr, va, splats=splats)[:4]
    for x, y in zip(got, want):
        assert torch.equal(x, y)
    # fresh records (a new projection into the same storage or not) carry nobody's slots
    splats2 = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], 1, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False, True,
                                        want_splats=True)[6]
    assert not ops._splat_slots_valid(tl_classic, splats2)

