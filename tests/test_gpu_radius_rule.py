"""The radius rule as a policy: radius_rule="opacity_aware" -- gsplat >= 1.5's per-axis extents
min(3.33, sqrt(2 ln(255 opacity))) sqrt(Sigma_ii), radii [C,N,2], Gaussians under 1/255 culled -- through every entry
point that projects (operator, fused per-camera path, mgs_render_frames, mgs_render_frames_train, FrameRenderer),
against the oracle restating the same rule (oracle/gs_oracle_np.py project(radius_rule=...)).  The default rule
("classic", A.2 step 5) is what every other test file exercises."""
import math

import numpy as np
import pytest
import torch

from oracle import gs_oracle_np as O
from oracle import gs_oracle_torch as OT
from robosimgs_amd import camera_ring, synthetic_scene

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _scene(n=10_000, mu=0.05, deg=0, w=256, h=256, theta=0.3, seed=0):
    g = synthetic_scene(n, math.log(mu), deg, seed)
    g.opacity_logits[::13] = -5.8                       # opacity 0.0030 < 1/255: the opacity-aware rule culls these
    g.opacity_logits[1::13] = -5.45                     # 0.0043: barely visible, extents of a fraction of a sigma
    cam = camera_ring(1, w, h, thetas=[theta])[0]
    return g, cam


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def _d(a, grad=False):
    return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=grad)


@pytest.mark.parametrize("n,mu,w,h,theta,aa,with_op", [(10_000, 0.05, 256, 256, 0.3, False, True),
                                                         (3_000, 0.2, 200, 120, 2.1, True, True),
                                                         (3_000, 0.2, 200, 120, 2.1, False, False)])
def test_projection_operator_per_axis_radii(n, mu, w, h, theta, aa, with_op):
    from robosimgs_amd import ops
    g, cam = _scene(n, mu, 0, w, h, theta)
    op = np.asarray(g.opacities, dtype=np.float32).copy()
    op[::11] = 0.0035                                   # under 1/255: the rule culls them
    ref = O.project(g.means, g.quats, g.scales, cam.viewmat(), cam.K, w, h, radius_rule="opacity_aware",
                    opacities=op.astype(np.float64) if with_op else None, antialiased=aa)
    radii, means2d, depths, conics, comps = ops.fully_fused_projection(
        _t(g.means), None, _t(g.quats), _t(g.scales), _t(cam.viewmat())[None], _t(cam.K)[None], w, h,
        calc_compensations=aa, opacities=_t(op) if with_op else None, radius_rule="opacity_aware")
    assert radii.shape == (1, n, 2) and radii.dtype == torch.int32
    radii = radii[0].cpu().numpy()
    vis_ref, vis = ref["radii"][:, 0] > 0, radii[:, 0] > 0
    assert ((radii[:, 0] > 0) == (radii[:, 1] > 0)).all()
    assert int((vis_ref != vis).sum()) <= max(1, n // 5000)
    both = vis_ref & vis
    dr = np.abs(radii[both] - ref["radii"][both])
    assert dr.max() <= 1 and (dr > 0).sum() <= max(2, n // 1000), f"extent mismatches: {(dr > 0).sum()} (max {dr.max()})"
    if with_op:
        assert not vis[::11].any()
    np.testing.assert_allclose(means2d[0].cpu().numpy()[both], ref["means2d"][both], rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(conics[0].cpu().numpy()[both], ref["conics"][both], rtol=2e-4, atol=1e-6)
    assert np.all(means2d[0].cpu().numpy()[~vis] == 0) and np.all(depths[0].cpu().numpy()[~vis] == 0)
    # the classic rule through the same entry point is unchanged: [C,N], ceil(3 sqrt(lambda_1))
    r0 = ops.fully_fused_projection(_t(g.means), None, _t(g.quats), _t(g.scales), _t(cam.viewmat())[None],
                                    _t(cam.K)[None], w, h)[0]
    assert r0.shape == (1, n)
    with pytest.raises(ValueError):
        ops.fully_fused_projection(_t(g.means), None, _t(g.quats), _t(g.scales), _t(cam.viewmat())[None],
                                   _t(cam.K)[None], w, h, radius_rule="inria")


@pytest.mark.parametrize("n,mu,w,h", [(10_000, 0.05, 256, 256), (2_000, 0.3, 200, 120), (5, 0.05, 256, 256)])
def test_isect_tiles_per_axis_radii_bit_exact(n, mu, w, h):
    """Integer path: per-axis radii [C,N,2] in, the lists of the stable sort on the rectangles mean +- (rx, ry) out."""
    from robosimgs_amd import ops
    g, cam = _scene(n, mu, 0, w, h)
    radii, means2d, depths, conics, _ = ops.fully_fused_projection(
        _t(g.means), None, _t(g.quats), _t(g.scales), _t(cam.viewmat())[None], _t(cam.K)[None], w, h,
        opacities=_t(g.opacities), radius_rule="opacity_aware")
    tw, th = -(-w // 16), -(-h // 16)
    tpg, isect_ids, flatten_ids = ops.isect_tiles(means2d, radii, depths, 16, tw, th)
    r_tpg, r_ids, r_flat = O.isect_tiles(means2d[0].cpu().numpy(), radii[0].cpu().numpy(), depths[0].cpu().numpy(),
                                         16, tw, th, dtype=np.float32)
    np.testing.assert_array_equal(tpg[0].cpu().numpy(), r_tpg)
    np.testing.assert_array_equal(isect_ids.cpu().numpy(), r_ids)
    np.testing.assert_array_equal(flatten_ids.cpu().numpy(), r_flat)
    # an anisotropic rectangle really is one: some Gaussian's x and y extents differ by a tile or more
    rr = radii[0].cpu().numpy()
    assert n < 100 or (np.abs(rr[:, 0] - rr[:, 1]) >= 16).any()


@pytest.mark.parametrize("mode,deg,aa", [("RGB", 0, False), ("RGB+ED", 3, False), ("RGB+D", 2, True)])
def test_rasterization_opacity_aware_matches_oracle(mode, deg, aa):
    """Whole path under the rule vs the whole oracle under the rule: the forward gate of every other forward test
    (zero unexplained pixels over 1e-4, flip-weight bound at could-flip pixels), classic and tight tile bounds,
    the per-camera path, the one-call inference path and FrameRenderer -- the same pixels bit for bit."""
    from robosimgs_amd import rasterization, FrameRenderer
    w, h = 256, 208
    g, cam = _scene(10_000, 0.05, deg, w, h)
    t = g.to_torch(DEV, deg)
    rm = "antialiased" if aa else "classic"
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    kw = dict(sh_degree=deg, render_mode=mode, rasterize_mode=rm, radius_rule="opacity_aware")
    colors, alphas, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, w, h,
                                         tile_bounds="classic", **kw)
    ref, ref_alpha, rmeta = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs[:, :(deg + 1) ** 2],
                                     _f32(cam.viewmat()), _f32(cam.K), w, h, sh_degree=deg, render_mode=mode,
                                     rasterize_mode=rm, margins=True, flip_eps=O.EPS_PATH, radius_rule="opacity_aware")
    assert meta["radii"].shape == (1, len(g), 2)
    n_vis = int(meta["radii"][0, :, 0].gt(0).sum())
    assert abs(n_vis - rmeta["n_vis"]) <= 1
    assert abs(int(meta["n_isects"][0]) - rmeta["n_isect"]) <= 8          # (an extent on a ceil knife edge moves a few pairs)
    st = O.check_frame(colors[0].cpu().numpy(), alphas[0].cpu().numpy(), ref, ref_alpha, rmeta["margins"],
                       O.EPS_PATH, rmeta["edge_mask"], expected_depth="E" in mode, what=f"opacity-aware {mode}",
                       flip_weight=rmeta["flip_weight"], feat_max=rmeta["feat_max"], require_flip_bound=True)
    print(f"\nopacity-aware rule {mode} deg {deg}: {st}; n_isect {int(meta['n_isects'][0])}")
    # against the classic rule: fewer pairs, and not the same image (the rule changes edge pixels, A.4)
    c0, a0, m0 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, w, h,
                               tile_bounds="classic", sh_degree=deg, render_mode=mode, rasterize_mode=rm)
    assert int(meta["n_isects"][0]) < 0.8 * int(m0["n_isects"][0])
    assert m0["radii"].shape == (1, len(g))
    assert not torch.equal(c0, colors)
    assert float((c0 - colors).abs().max()) < 0.05
    # tightened rectangles on top of the rule: same pixels bit for bit, lists no longer
    c2, a2, m2 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, w, h, **kw)
    assert torch.equal(c2, colors) and torch.equal(a2, alphas)
    assert int(m2["n_isects"][0]) <= int(meta["n_isects"][0])
    assert torch.equal(m2["radii"], meta["radii"])
    # one C call per batch of inference frames (mgs_render_frames) and FrameRenderer's graphs
    c3, a3, m3 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, w, h,
                               isect_capacity=400_000, lean_meta=True, **kw)
    assert torch.equal(c3, colors) and torch.equal(a3, alphas) and int(m3["n_isects"][0]) == int(m2["n_isects"][0])
    fr = FrameRenderer(t, w, h, render_mode=mode, isect_capacity=400_000, frames_in_flight=2, reorder=None,
                       rasterize_mode=rm, radius_rule="opacity_aware")
    f = fr.fetch(fr.submit(cam.viewmat(), cam.K))
    assert torch.equal(f["colors"], colors[0]) and torch.equal(f["alphas"], alphas[0])


@pytest.mark.parametrize("deg,mode,aa,cap", [(2, "RGB+ED", False, None), (1, "RGB", True, None), (3, "RGB", False, 300_000)])
def test_rasterization_opacity_aware_backward(deg, mode, aa, cap):
    """Gradients under the rule vs fp64 autograd of the torch oracle under the rule (the extent is not differentiable:
    only the visible set and the lists change).  cap given: the batched training path (mgs_render_frames_train /
    _backward), whose state carries radii_y.  The gate is the classic rule's: cosine >= 0.999, <= 1 % of the rows over
    5e-3 and -- the fp64 port restates the rule too -- EVERY row within rounding + 1.5 x its flip budget (the
    anti-aliased case keeps the fraction bound: the port has no anti-aliased mode)."""
    from robosimgs_amd import rasterization
    from grad_gate import compare
    w, h = 112, 80
    g, cam = _scene(6000, 0.07, deg, w, h)
    t = g.to_torch(DEV, deg)
    names = ["means", "quats", "scales", "opacities", "colors"]
    for k in names:
        t[k].requires_grad_(True)
    rm = "antialiased" if aa else "classic"
    colors, alphas, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                                         _t(cam.viewmat()[None]), _t(cam.K[None]), w, h, sh_degree=deg, render_mode=mode,
                                         rasterize_mode=rm, radius_rule="opacity_aware", isect_capacity=cap)
    assert meta["radii"].shape == (1, len(g), 2)
    rng = np.random.default_rng(2)
    wr, wa = rng.normal(size=tuple(colors.shape[1:])), rng.normal(size=(h, w))
    ((colors[0] * _t(wr)).sum() + (alphas[0, ..., 0] * _t(wa)).sum()).backward()
    r = {k: _d(v, True) for k, v in (("means", g.means), ("quats", g.quats), ("scales", g.scales),
                                      ("opacities", g.opacities), ("colors", g.sh_coeffs[:, :(deg + 1) ** 2]))}
    img, al, p = OT.render(r["means"], r["quats"], r["scales"], r["opacities"], r["colors"], _d(cam.viewmat()), _d(cam.K),
                           w, h, sh_degree=deg, render_mode=mode, rasterize_mode=rm, radius_rule="opacity_aware")
    ((img * _d(wr)).sum() + (al[..., 0] * _d(wa)).sum()).backward()
    rr = meta["radii"][0].cpu().numpy()
    assert int(((rr[:, 0] > 0) != (p["radii"][:, 0].numpy() > 0)).sum()) <= 1
    budgets = {k: None for k in names}
    if not aa:
        from grad_gate import oracle_budgets, parameter_budgets
        f32 = lambda m: np.asarray(m, dtype=np.float32)
        info = oracle_budgets(g, f32(cam.viewmat()), f32(cam.K), w, h, deg, mode, wr, wa, O.EPS_PATH_GRAD,
                              radius_rule="opacity_aware")
        bud = info["budget"]
        assert ((info["radii"] > 0) == (rr[:, 0] > 0)).sum() >= len(g) - 1
        budgets.update(parameter_budgets(g, f32(cam.viewmat()), f32(cam.K), w, h, deg, mode != "RGB", bud,
                                         radius_rule="opacity_aware"))
        budgets["opacities"] = bud[:, 3]
    for k in names:
        ref = r[k].grad.numpy()
        compare("opacity-aware v_" + k, t[k].grad, ref if ref.ndim > 1 else ref.reshape(-1, 1), row_tol=5e-3,
                bad_frac=1e-2, cos_min=0.999, budget=budgets[k])
    # culled by the rule (opacity < 1/255) = no gradient
    culled = rr[:, 0] == 0
    assert culled.any() and float(t["means"].grad[torch.from_numpy(culled).to(DEV)].abs().max()) == 0.0
