"""BASELINE.json's full-size configurations on the GPU: direct comparison with the C++ port
(oracle/gs_cpu.cpp; it finishes in seconds on the GPU box's host cores) and size-independent
properties of the path: sorted / complete tile lists, linearity in the colour features,
camera-order equivariance, alpha range."""
import math

import numpy as np
import pytest
import torch

from oracle import cpu_ref
from oracle import gs_oracle_np as O
from robosimgs_amd import camera_ring, synthetic_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


@pytest.fixture(scope="module")
def config2():
    g = synthetic_scene(1_000_000, math.log(0.012), 3, 0)
    cam = camera_ring(1, 1920, 1080, thetas=[0.3])[0]
    return g, cam, g.to_torch(DEV, 3)


def test_config2_matches_cpu_port_and_survey_counts(config2):
    """configs[1]: 1 M Gaussians, SH 3, 1920x1080 forward."""
    from robosimgs_amd import rasterization
    g, cam, t = config2
    c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                               _t(cam.viewmat())[None], _t(cam.K)[None], 1920, 1080, sh_degree=3,
                               tile_bounds="classic")
    ct, at, mt = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                               _t(cam.viewmat())[None], _t(cam.K)[None], 1920, 1080, sh_degree=3)
    assert torch.equal(ct, c) and torch.equal(at, a)                 # tight tile bounds: same bits
    assert int(mt["n_isects"][0]) < 0.8 * int(meta["n_isects"][0])
    n_isect = int(meta["n_isects"][0])
    assert int((meta["radii"] > 0).sum()) == 764_945                 # SURVEY.md 8(d) calibration
    assert abs(n_isect - 5_019_708) <= 100                           # fp32 vs fp64 knife edges
    # the fp64 instantiation of the C++ port is the reference answer; it also says where a branch of
    # the blend (or a knife edge of the projection) came within O.EPS_PATH of flipping.  Every pixel
    # over the north-star tolerance must be one of those: zero unexplained pixels out of 2 M.
    ref, ra, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs,
                                       cam.viewmat(), cam.K, 1920, 1080, 3, flip_eps=O.EPS_PATH)
    assert abs(info["n_isect"] - 5_019_708) <= 2 and abs(info["n_isect"] - n_isect) <= 100   # (fp32-rounded camera)
    st = O.check_frame(c[0].cpu().numpy(), a[0].cpu().numpy(), ref, ra, info["margins"], O.EPS_PATH,
                       info["edge_mask"], what="configs[1]", flip_weight=info["flip_weight"], feat_max=info["feat_max"],
                       require_flip_bound=True)
    print(f"\nconfigs[1] vs fp64 port: {st}, knife-edge Gaussians {info['n_edge_gaussians']}")
    al = a[0, ..., 0]
    assert float(al.min()) >= 0.0 and float(al.max()) < 1.0


def test_config2_opacity_aware_radius_rule_matches_fp64_port(config2):
    """configs[1] under the other radius policy (rasterization(radius_rule="opacity_aware")): the same
    zero-unexplained-pixel gate against the fp64 port restating that rule, at full size; tightened rectangles on top of
    it change no bit; what the rule does to the work (2,106 Gaussians under 1/255 culled, 18 % fewer pairs than gsplat
    1.4's squares, a different image where opaque Gaussians reach beyond 3 sigma)."""
    from robosimgs_amd import rasterization
    g, cam, t = config2
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    kw = dict(sh_degree=3, render_mode="RGB+ED", radius_rule="opacity_aware")
    c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, 1920, 1080,
                               tile_bounds="classic", **kw)
    ct, at, mt = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, 1920, 1080, **kw)
    assert torch.equal(ct, c) and torch.equal(at, a)
    assert meta["radii"].shape == (1, 1_000_000, 2)
    n_vis, n_isect = int((meta["radii"][0, :, 0] > 0).sum()), int(meta["n_isects"][0])
    assert abs(n_vis - 762_839) <= 2 and abs(n_isect - 4_105_508) <= 200 and int(mt["n_isects"][0]) <= n_isect
    ref, ra, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, cam.viewmat(), cam.K,
                                       1920, 1080, 3, with_depth=True, flip_eps=O.EPS_PATH, radius_rule="opacity_aware")
    assert abs(info["n_isect"] - n_isect) <= 200 and abs(info["n_vis"] - n_vis) <= 2
    ref = ref.astype(np.float64)
    ref[..., 3] /= np.maximum(ra, 1e-10)                              # the port returns the depth sum ("D")
    st = O.check_frame(c[0].cpu().numpy(), a[0].cpu().numpy(), ref, ra, info["margins"], O.EPS_PATH, info["edge_mask"],
                       expected_depth=True, what="configs[1], opacity-aware rule", flip_weight=info["flip_weight"],
                       feat_max=info["feat_max"], require_flip_bound=True)
    print(f"\nconfigs[1], opacity-aware radius rule vs fp64 port: {st}, knife-edge Gaussians {info['n_edge_gaussians']}")
    c0, _, m0 = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, 1920, 1080,
                              sh_degree=3, render_mode="RGB+ED", tile_bounds="classic")
    assert n_isect < 0.85 * int(m0["n_isects"][0]) and not torch.equal(c0, c)


def test_config2_headline_path_frame_renderer_in_morton_order_matches_fp64_port(config2):
    """The path bench.py's `value` is measured on: FrameRenderer with three frames in flight, its resident copy of the
    scene in Morton order, the per-tile raster schedule, "RGB+ED", lean frames through mgs_render_frames -- every frame of
    a five-camera sweep (the bench camera and four ring cameras) against the fp64 port on the scene in the CALLER's order,
    zero unexplained pixels and the flip bound.  (Depth ties resolve by the copy's indices: they are could-flip pixels of
    the depth-order margin.)"""
    from robosimgs_amd import FrameRenderer
    g, cam, t = config2
    W, H = 1920, 1080
    cams = [cam] + list(camera_ring(4, W, H, thetas=[1.1, 2.6, 4.0, 5.5]))
    fr = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=3, sizing_camera=(cam.viewmat(), cam.K),
                       capacity_margin=1.6)
    assert fr.order is not None and fr.kw["raster_schedule"] == "throughput" and fr.kw["lean_meta"]
    frames = {}
    fr.render_sequence(cams, lambda i, f: frames.__setitem__(i, (f["colors"].cpu().numpy(), f["alphas"].cpu().numpy())))
    for i, c in enumerate(cams):
        ref, ra, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, c.viewmat(), c.K, W, H, 3,
                                           with_depth=True, flip_eps=O.EPS_PATH)
        ref = ref.astype(np.float64)
        ref[..., 3] /= np.maximum(ra, 1e-10)                  # the port returns the depth sum ("D")
        st = O.check_frame(frames[i][0], frames[i][1], ref, ra, info["margins"], O.EPS_PATH, info["edge_mask"],
                           expected_depth=True, what=f"FrameRenderer camera {i}", flip_weight=info["flip_weight"],
                           feat_max=info["feat_max"], require_flip_bound=True)
        print(f"\nFrameRenderer (Morton copy, 3 in flight) camera {i}: {st}")


def test_config2_tile_lists_sorted_and_complete(config2):
    from robosimgs_amd import ops
    g, cam, t = config2
    radii, m2d, dep, con, _, feats = ops.project_color_fwd_raw(
        t["means"], t["quats"], t["scales"], t["opacities"], 3, t["colors"], _t(cam.viewmat()),
        _t(cam.K), 1920, 1080, 0.3, 0.01, 1e10, 0.0, False, False)
    tl = ops.isect_tiles_raw(m2d, radii, dep, 120, 68, 6_500_000, want_isect_ids=True,
                             want_pair_info=True)
    n = int(tl.n_isect.item())
    assert int(tl.status.item()) == 0 and n == int(tl.tiles_per_gauss.sum().item())
    keys = tl.isect_ids[:n]
    assert bool((keys[1:] >= keys[:-1]).all())                        # globally sorted by (tile, depth)
    ids = tl.flatten_ids[:n].long()
    tiles = (keys >> 32)
    assert bool((tiles == tl.tile_ids[:n].long()).all())
    assert bool(((keys & 0xffffffff) == dep[ids].view(torch.int32).long()).all())
    # ties in (tile, depth) keep Gaussian-index order
    same = keys[1:] == keys[:-1]
    assert bool((ids[1:][same] > ids[:-1][same]).all())
    # offsets are the first index of each tile, checksum of ids is order independent
    offs = tl.tile_offsets.long()
    assert int(offs[-1]) == n and bool((offs[1:] >= offs[:-1]).all())
    counts = torch.bincount(tiles, minlength=120 * 68)
    assert bool((counts == (offs[1:] - offs[:-1])).all())
    per_gauss = torch.bincount(ids, minlength=len(g))
    assert bool((per_gauss == tl.tiles_per_gauss.long()).all())


def test_linearity_and_camera_equivariance():
    """Blend is linear in the colour features; a camera batch equals the cameras one by one."""
    from robosimgs_amd import rasterization
    g = synthetic_scene(200_000, math.log(0.02), 0, 5)
    cams = camera_ring(4, 640, 360)
    t = g.to_torch(DEV, 0)
    vm = _t(np.stack([c.viewmat() for c in cams]))
    Ks = _t(np.stack([c.K for c in cams]))
    fa, fb = torch.rand(len(g), 5, device=DEV), torch.rand(len(g), 5, device=DEV)
    args = (t["means"], t["quats"], t["scales"], t["opacities"])
    ra, al, _ = rasterization(*args, fa, vm, Ks, 640, 360)
    rb, _, _ = rasterization(*args, fb, vm, Ks, 640, 360)
    rab, _, _ = rasterization(*args, 2.0 * fa - 0.5 * fb, vm, Ks, 640, 360)
    torch.testing.assert_close(rab, 2.0 * ra - 0.5 * rb, rtol=1e-4, atol=2e-5)
    perm = [2, 0, 3, 1]
    rp, ap, _ = rasterization(*args, fa, vm[perm], Ks[perm], 640, 360)
    assert torch.equal(rp, ra[perm]) and torch.equal(ap, al[perm])


@pytest.mark.timeout(600)
def test_config5_stress_matches_cpu_port():
    """configs[4]: 5 M Gaussians, SH 3, 3840x2160 (mean 1,105 Gaussians per tile)."""
    from robosimgs_amd import rasterization, check_isect_status
    g = synthetic_scene(5_000_000, math.log(0.008), 3, 0)
    cam = camera_ring(1, 3840, 2160, thetas=[0.3])[0]
    t = g.to_torch(DEV, 3)
    c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                               _t(cam.viewmat())[None], _t(cam.K)[None], 3840, 2160, sh_degree=3,
                               isect_capacity=40_000_000, tile_bounds="classic")
    check_isect_status(meta)
    ct, at, mt = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                               _t(cam.viewmat())[None], _t(cam.K)[None], 3840, 2160, sh_degree=3,
                               isect_capacity=40_000_000)
    assert torch.equal(ct, c) and torch.equal(at, a)
    assert int(mt["n_isects"][0]) < int(meta["n_isects"][0])
    del ct, at, mt
    assert int((meta["radii"] > 0).sum()) == 3_797_688
    assert abs(int(meta["n_isects"][0]) - 35_799_376) <= 600
    ref, ra, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs,
                                       cam.viewmat(), cam.K, 3840, 2160, 3, flip_eps=O.EPS_PATH)
    assert abs(info["n_isect"] - 35_799_376) <= 20
    st = O.check_frame(c[0].cpu().numpy(), a[0].cpu().numpy(), ref, ra, info["margins"], O.EPS_PATH,
                       info["edge_mask"], what="configs[4]", flip_weight=info["flip_weight"], feat_max=info["feat_max"],
                       require_flip_bound=True)
    print(f"\nconfigs[4] vs fp64 port: {st}, knife-edge Gaussians {info['n_edge_gaussians']}")


def test_config3_backward_matches_fp64_oracle(config2):
    """configs[2] at full size (1 M Gaussians, 1920x1080, forward + backward): gradients of
    L = <w, render> + <u, alpha> against the fp64 oracle chain -- A.2 step 10 in the C++ port
    (oracle/gs_cpu.cpp, fp64 sums, validated against autograd in tests/test_oracle_cpu.py) for the
    blend, then torch autograd (fp64, oracle/gs_oracle_torch.py) through projection and SH colour.
    Tolerance: per-row error scaled by the row's magnitude + 1e-3 of the tensor's largest entry stays
    under 5e-3 on all but 1 % of the rows, cosine of the whole gradient >= 0.999 (as at test size)."""
    from robosimgs_amd import rasterization
    from oracle import gs_oracle_torch as OT
    from test_gpu_backward import _compare
    g, cam, t = config2
    W, H, deg = 1920, 1080, 3
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    rng = np.random.default_rng(21)
    w_img = rng.normal(size=(H, W, 3)).astype(np.float32)
    w_a = rng.normal(size=(H, W)).astype(np.float32)
    names = ("means", "quats", "scales", "opacities", "colors")
    p = {k: t[k].clone().requires_grad_(True) for k in names}
    c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, W, H,
                               sh_degree=deg)
    ((c[0] * _t(w_img)).sum() + (a[0, ..., 0] * _t(w_a)).sum()).backward()
    got = {k: p[k].grad.detach().cpu() for k in names}
    got_blend = [x.detach().cpu() for x in meta["blend_grads"][0]]
    del c, a, meta, p
    # oracle: blend backward in the fp64 port ...
    vmf, Kf = np.asarray(cam.viewmat(), np.float32), np.asarray(cam.K, np.float32)
    _, _, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vmf, Kf, W, H, deg,
                                    margins=True, v_render=w_img, v_alpha=w_a, want_projected=True,
                                    flip_eps=O.EPS_PATH_GRAD, want_touched=True, want_budget=True)
    vis = info["radii"] > 0
    # EVERY row within rounding + 1.5 x its flip budget of the oracle's (tests/grad_gate.py; oracle/gs_cpu.cpp
    # Extras::budget): the blend stage's four outputs first ...
    bud = info["budget"]
    print(f"\nGaussians that reach a could-flip pixel: {info['touched'][vis].mean():.1%} of the visible ones")
    for name, got_b, ref_b, b in (("means2d", got_blend[0], info["g_means2d"], bud[:, 0]),
                                  ("conics", got_blend[1], info["g_conics"], bud[:, 1]),
                                  ("feats", got_blend[2], info["g_feats"], bud[:, 2]),
                                  ("opacities", got_blend[3], info["g_opacities"].reshape(-1, 1), bud[:, 3])):
        _compare(f"v_{name} (blend)", got_b, ref_b, row_tol=5e-3, bad_frac=1e-2, cos_min=0.999, budget=b)
    _compare("v_opacities", got["opacities"], info["g_opacities"].reshape(-1, 1), row_tol=5e-3, bad_frac=1e-2,
             cos_min=0.999, budget=bud[:, 3])
    # ... then autograd through projection + SH colour, vectorised over the 1 M Gaussians (fp64, CPU)
    d = lambda x, grad=False: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=grad)
    r = {"means": d(g.means, True), "quats": d(g.quats, True), "scales": d(g.scales, True),
         "colors": d(g.sh_coeffs[:, :(deg + 1) ** 2], True)}
    vmt, Kt = d(vmf), d(Kf)
    pr = OT.project(r["means"], r["quats"], r["scales"], vmt, Kt, W, H)
    assert int((pr["radii"] > 0).sum()) == int(vis.sum())
    campos = -vmt[:3, :3].T @ vmt[:3, 3]
    rgb = torch.clamp(OT.spherical_harmonics(deg, r["means"] - campos, r["colors"]) + 0.5, min=0.0)
    rgb = rgb * torch.tensor(vis, dtype=torch.float64)[:, None]
    np.testing.assert_allclose(pr["means2d"].detach().numpy(), info["means2d"], atol=1e-6)
    # (... the budgets of the parameter rows first: absolute Jacobian of the same graph, one pass per output component)
    from grad_gate import chained_budget
    pb = chained_budget(r, {"means2d": pr["means2d"], "conics": pr["conics"], "feats": rgb},
                        {"means2d": bud[:, 0], "conics": bud[:, 1], "feats": bud[:, 2]})
    (pr["means2d"] * d(info["g_means2d"])).sum().add((pr["conics"] * d(info["g_conics"])).sum()) \
        .add((rgb * d(info["g_feats"])).sum()).backward()
    for k in ("means", "quats", "scales", "colors"):
        ref = r[k].grad.numpy()
        _compare("v_" + k, got[k], ref.reshape(ref.shape[0], -1), row_tol=5e-3, bad_frac=1e-2, cos_min=0.999,
                 budget=pb[k].reshape(ref.shape[0], -1))


def _timed_step_gate(g, cam, t, cap, label, W=1920, H=1080, deg=3):
    """One training step exactly as bench.py's `fwd_bwd` leg times it -- render_mode "RGB+ED" (four channels, the
    expected-depth divide undone in the backward's prologue), the fused L1 loss to the seed-1 U(0,1) target on all four
    channels (its cotangent is sign(render - target) / n), a fixed list capacity, i.e. the batched training calls
    mgs_render_frames_train / mgs_render_frames_backward, the segmented backward walk (backward_segment 256) -- against the
    fp64 oracle chain: every row of the nine gradient tensors within rounding + 1.5 x its flip budget (tests/grad_gate.py;
    blend: oracle/gs_cpu.cpp, then fp64 autograd through projection, SH and depth)."""
    from robosimgs_amd import rasterization, l1_loss
    from oracle import gs_oracle_torch as OT
    from grad_gate import compare, oracle_budgets, chained_budget
    mode = "RGB+ED"
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    names = ("means", "quats", "scales", "opacities", "colors")
    target = torch.rand(1, H, W, 4, device=DEV, generator=torch.Generator(DEV).manual_seed(1))      # bench.py: bench_fwd_bwd

    def step(cap_):
        p = {k: t[k].detach().clone().requires_grad_(True) for k in names}
        c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, W, H,
                                   sh_degree=deg, render_mode=mode, isect_capacity=cap_)
        l1_loss(c, target).backward()
        return c.detach(), {k: p[k].grad.detach() for k in names}, meta

    # the timed form (fixed capacity: one HIP-graph-capturable step) and the per-camera entry points, which also publish
    # the blend stage's own four gradients; the two must agree bit for bit
    c_cap, got, meta_cap = step(cap)
    assert int(meta_cap["isect_status"].max()) == 0
    c_pc, got_pc, meta = step(None)
    assert torch.equal(c_cap, c_pc)
    for k in names:
        assert torch.equal(got[k], got_pc[k]), k
    got_blend = [x.detach().cpu() for x in meta["blend_grads"][0]]
    assert torch.equal(meta_cap["means2d_grad"][0].cpu(), got_blend[0])
    w_img = (torch.sign(c_cap[0] - target[0]) / float(c_cap.numel())).cpu().numpy()      # what mgs_l1_loss_fwd_grad leaves
    got = {k: v.cpu() for k, v in got.items()}
    del c_cap, c_pc, got_pc, meta, meta_cap
    vmf, Kf = np.asarray(cam.viewmat(), np.float32), np.asarray(cam.K, np.float32)
    info = oracle_budgets(g, vmf, Kf, W, H, deg, mode, w_img, np.zeros((H, W), np.float32), O.EPS_PATH_GRAD)
    vis, bud = info["radii"] > 0, info["budget"]
    print(f"\n{label}: Gaussians that reach a could-flip pixel: {info['touched'][vis].mean():.1%} of the visible ones")
    for name, got_b, ref_b, b in (("means2d", got_blend[0], info["g_means2d"], bud[:, 0]),
                                  ("conics", got_blend[1], info["g_conics"], bud[:, 1]),
                                  ("feats", got_blend[2], info["g_feats"], bud[:, 2]),
                                  ("opacities", got_blend[3], info["g_opacities"].reshape(-1, 1), bud[:, 3])):
        compare(f"{label} v_{name} (blend)", got_b, ref_b, row_tol=5e-3, bad_frac=1e-2, cos_min=0.999, budget=b)
    compare(f"{label} v_opacities", got["opacities"], info["g_opacities"].reshape(-1, 1), row_tol=5e-3, bad_frac=1e-2,
            cos_min=0.999, budget=bud[:, 3])
    d = lambda x, grad=False: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=grad)
    r = {"means": d(g.means, True), "quats": d(g.quats, True), "scales": d(g.scales, True),
         "colors": d(g.sh_coeffs[:, :(deg + 1) ** 2], True)}
    vmt, Kt = d(vmf), d(Kf)
    pr = OT.project(r["means"], r["quats"], r["scales"], vmt, Kt, W, H)
    assert int((pr["radii"] > 0).sum()) == int(vis.sum())
    campos = -vmt[:3, :3].T @ vmt[:3, 3]
    rgb = torch.clamp(OT.spherical_harmonics(deg, r["means"] - campos, r["colors"]) + 0.5, min=0.0)
    rgb = rgb * torch.tensor(vis, dtype=torch.float64)[:, None]
    feats = torch.cat([rgb, pr["depths"][:, None]], dim=-1)            # the fourth blended channel is the camera depth
    pb = chained_budget(r, {"means2d": pr["means2d"], "conics": pr["conics"], "feats": feats},
                        {"means2d": bud[:, 0], "conics": bud[:, 1], "feats": bud[:, 2]})
    (pr["means2d"] * d(info["g_means2d"])).sum().add((pr["conics"] * d(info["g_conics"])).sum()) \
        .add((feats * d(info["g_feats"])).sum()).backward()
    for k in ("means", "quats", "scales", "colors"):
        ref = r[k].grad.numpy()
        compare(f"{label} v_" + k, got[k], ref.reshape(ref.shape[0], -1), row_tol=5e-3, bad_frac=1e-2, cos_min=0.999,
                budget=pb[k].reshape(ref.shape[0], -1))


def test_config3_timed_training_step_rgb_ed_l1_matches_fp64_oracle(config2):
    """configs[2] exactly as bench.py's `fwd_bwd` leg times it (see _timed_step_gate): 0 of 1 M rows over budget in all
    nine gradient tensors."""
    g, cam, t = config2
    _timed_step_gate(g, cam, t, 4_700_000, "RGB+ED L1")


# ---- a scene shaped like an export (NOT a BASELINE.json config): clustered means, heavy-tailed extents, a handful of
# ---- screen-filling Gaussians, thousands of needle-like ones (robosimgs_amd.synthetic_scene_heavy_tailed) --------------
@pytest.fixture(scope="module")
def heavy():
    from robosimgs_amd import synthetic_scene_heavy_tailed
    g = synthetic_scene_heavy_tailed(1_000_000, sh_degree=3, seed=0)
    cam = camera_ring(1, 1920, 1080, thetas=[0.3])[0]
    return g, cam, g.to_torch(DEV, 3)


@pytest.mark.timeout(900)
def test_heavy_tailed_scene_forward_is_closer_to_fp64_than_a_plain_fp32_restatement(heavy):
    """1 M Gaussians at 1920x1080 with lists of 41 ... 39 k entries (mean 933; SURVEY 8(d)'s scene: 615 everywhere), ten
    Gaussians that cover every tile, 5.6 k over 196 tiles, 4,000 needle-like ones.  Such a scene is ill-conditioned for
    fp32: the C++ port's FLOAT instantiation -- the reference's formulas, plainly, in fp32 -- is 589 pixels over 1e-4 from
    the fp64 answer, 488 of them at no near-flip decision (the HIP path: 315 / 226).  The gate that can hold (and the claim that matters: no
    cliff) is oracle.gs_oracle_np.check_frame_against_fp32_port: the HIP path no farther from fp64 than that restatement,
    by count, by unexplained count, by the error's high percentiles and maximum; classic and tightened rectangles
    bit-identical; the headline path (FrameRenderer, Morton copy, three in flight, per-tile schedule) through the same."""
    from robosimgs_amd import rasterization, FrameRenderer
    g, cam, t = heavy
    W, H = 1920, 1080
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    kw = dict(sh_degree=3, render_mode="RGB+ED")
    c, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, W, H,
                               tile_bounds="classic", **kw)
    ct, at, mt = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm, K, W, H, **kw)
    assert torch.equal(ct, c) and torch.equal(at, a)
    n_isect, n_tight = int(meta["n_isects"][0]), int(mt["n_isects"][0])
    lens = (mt["tile_lists"][0].tile_offsets[1:] - mt["tile_lists"][0].tile_offsets[:-1])
    tpg = meta["tiles_per_gauss"][0]
    print(f"\nheavy-tailed scene: n_isect {n_isect} classic / {n_tight} tightened; list length mean {float(lens.float().mean()):.0f} "
          f"max {int(lens.max())}; tiles per Gaussian max {int(tpg.max())}, over 196 tiles: {int((tpg > 196).sum())}")
    assert int(tpg.max()) == 120 * 68 and int(lens.max()) > 20_000 and n_tight < n_isect
    ref, ra, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, cam.viewmat(), cam.K, W, H, 3,
                                       with_depth=True, flip_eps=O.EPS_PATH)
    assert abs(info["n_isect"] - n_isect) <= 400
    ref = ref.astype(np.float64)
    ref[..., 3] /= np.maximum(ra, 1e-10)                              # the port returns the depth sum ("D")
    r32, a32, _ = cpu_ref.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, np.asarray(cam.viewmat(), np.float32),
                                 np.asarray(cam.K, np.float32), W, H, 3, with_depth=True)
    st = O.check_frame_against_fp32_port(c[0].cpu().numpy(), a[0].cpu().numpy(), ref, ra, r32, a32, info["margins"], O.EPS_PATH,
                                         info["edge_mask"], expected_depth=True, what="heavy-tailed scene")
    print(f"heavy-tailed scene vs fp64 port (and the fp32 port beside it): {st}")
    fr = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=3, sizing_camera=(cam.viewmat(), cam.K),
                       capacity_margin=1.3)
    f = fr.render(cam.viewmat(), cam.K)
    st = O.check_frame_against_fp32_port(f["colors"].cpu().numpy(), f["alphas"].cpu().numpy(), ref, ra, r32, a32, info["margins"],
                                         O.EPS_PATH, info["edge_mask"], expected_depth=True, what="heavy-tailed scene, FrameRenderer")
    print(f"heavy-tailed scene, FrameRenderer (Morton copy, per-tile schedule): {st}")


@pytest.mark.timeout(900)
def test_heavy_tailed_scene_training_step_gradients(heavy):
    """The timed training step (RGB+ED, L1 cotangent, fixed capacity = the batched calls, segments of 256) on the
    heavy-tailed scene: lists of up to 39 k entries are 150 segments of the backward's walk, the screen-filling Gaussians own
    8,160 record slots each.  Fixed-capacity and per-camera paths bit-identical; gradients against the fp64 oracle chain
    with the fraction rule (at most 1 % of the rows over the row tolerance, cosine >= 0.999) AND the per-row flip budgets
    asserted with a stated exemption: at most MAX_EXEMPT of the 1 M rows of a tensor may exceed rounding + 1.5 x budget
    (round 5 printed the count: 1) -- on this scene the projection's fp32 noise is part of the whole path and no budget
    prices it; the blend STAGE is held to its budgets without exemption (tests/test_gpu_heavy.py)."""
    MAX_EXEMPT = 8
    from robosimgs_amd import rasterization, l1_loss
    from oracle import gs_oracle_torch as OT
    from grad_gate import compare, oracle_budgets
    g, cam, t = heavy
    W, H, deg, mode = 1920, 1080, 3, "RGB+ED"
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    names = ("means", "quats", "scales", "opacities", "colors")
    target = torch.rand(1, H, W, 4, device=DEV, generator=torch.Generator(DEV).manual_seed(1))

    def step(cap_):
        p = {k: t[k].detach().clone().requires_grad_(True) for k in names}
        c, a, meta = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K, W, H,
                                   sh_degree=deg, render_mode=mode, isect_capacity=cap_)
        l1_loss(c, target).backward()
        return c.detach(), {k: p[k].grad.detach() for k in names}, meta

    c_cap, got, meta_cap = step(7_500_000)
    assert int(meta_cap["isect_status"].max()) == 0
    c_pc, got_pc, meta = step(None)
    assert torch.equal(c_cap, c_pc)
    for k in names:
        assert torch.equal(got[k], got_pc[k]) and bool(torch.isfinite(got[k]).all()), k
    got_blend = [x.detach().cpu() for x in meta["blend_grads"][0]]
    w_img = (torch.sign(c_cap[0] - target[0]) / float(c_cap.numel())).cpu().numpy()
    got = {k: v.cpu() for k, v in got.items()}
    del c_cap, c_pc, got_pc, meta, meta_cap
    vmf, Kf = np.asarray(cam.viewmat(), np.float32), np.asarray(cam.K, np.float32)
    info = oracle_budgets(g, vmf, Kf, W, H, deg, mode, w_img, np.zeros((H, W), np.float32), O.EPS_PATH_GRAD)
    bud = info["budget"]
    for name, got_b, ref_b, b in (("means2d", got_blend[0], info["g_means2d"], bud[:, 0]),
                                  ("conics", got_blend[1], info["g_conics"], bud[:, 1]),
                                  ("feats", got_blend[2], info["g_feats"], bud[:, 2]),
                                  ("opacities", got_blend[3], info["g_opacities"].reshape(-1, 1), bud[:, 3])):
        st = compare(f"heavy-tailed v_{name} (blend)", got_b, ref_b, row_tol=5e-3, bad_frac=1e-2, cos_min=0.999, verbose=False)
        # (budget statistics, for the record)
        gb = np.asarray(got_b, np.float64).reshape(len(b), -1)
        rb = np.asarray(ref_b, np.float64).reshape(len(b), -1)
        scale = np.abs(rb).max(axis=1, keepdims=True) + 1e-3 * np.abs(rb).max() + 1e-30
        over = (np.abs(gb - rb) / (2.0 * 5e-3 * scale + 1.5 * b.reshape(-1, 1))).max(axis=1) > 1.0
        print(f"\nheavy-tailed v_{name} (blend): {st}; rows over rounding + 1.5 x flip budget: {int(over.sum())} {np.flatnonzero(over)[:8].tolist()}")
        # the WHOLE path on this scene carries the projection's fp32 noise into sigma (the forward test's docstring), which no
        # flip budget prices; the blend stage on its own inputs is held to its budgets row by row
        # (tests/test_gpu_heavy.py::test_full_size_heavy_tailed_blend_stage_backward: 0 of 1 M).  Here: at most
        # MAX_EXEMPT rows per tensor may exceed theirs.
        assert int(over.sum()) <= MAX_EXEMPT, f"heavy-tailed v_{name}: {int(over.sum())} rows over rounding + 1.5 x their flip budget (allowed: {MAX_EXEMPT})"
    compare("heavy-tailed v_opacities", got["opacities"], info["g_opacities"].reshape(-1, 1), row_tol=5e-3, bad_frac=1e-2,
            cos_min=0.999, verbose=False)
    d = lambda x, grad=False: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=grad)
    r = {"means": d(g.means, True), "quats": d(g.quats, True), "scales": d(g.scales, True),
         "colors": d(g.sh_coeffs[:, :(deg + 1) ** 2], True)}
    vmt, Kt = d(vmf), d(Kf)
    pr = OT.project(r["means"], r["quats"], r["scales"], vmt, Kt, W, H)
    vis = info["radii"] > 0
    campos = -vmt[:3, :3].T @ vmt[:3, 3]
    rgb = torch.clamp(OT.spherical_harmonics(deg, r["means"] - campos, r["colors"]) + 0.5, min=0.0)
    rgb = rgb * torch.tensor(vis, dtype=torch.float64)[:, None]
    feats = torch.cat([rgb, pr["depths"][:, None]], dim=-1)
    (pr["means2d"] * d(info["g_means2d"])).sum().add((pr["conics"] * d(info["g_conics"])).sum()) \
        .add((feats * d(info["g_feats"])).sum()).backward()
    for k in ("means", "quats", "scales", "colors"):
        ref = r[k].grad.numpy()
        st = compare("heavy-tailed v_" + k, got[k], ref.reshape(ref.shape[0], -1), row_tol=5e-3, bad_frac=1e-2, cos_min=0.999,
                     verbose=False)
        print(f"heavy-tailed v_{k}: {st}")


def test_config4_block_of_eight_ring_cameras_through_render_sharded(config2):
    """configs[3]'s per-GPU share at its real shape: 8 consecutive cameras of the 64-camera ring at
    1920x1080 through `render_sharded(renderer=FrameRenderer)` (world of one, no collective); two of
    the frames are checked against the fp64 port with the zero-unexplained-pixels gate."""
    from robosimgs_amd import FrameRenderer
    from robosimgs_amd.distributed import render_sharded, shard_cameras
    g, cam, t = config2
    W, H = 1920, 1080
    block = shard_cameras(64, 8, 3)                          # rank 3 of 8: cameras 24..31
    assert list(block) == list(range(24, 32))
    cams = camera_ring(len(block), W, H, thetas=[2.0 * math.pi * k / 64 for k in block])
    vms = _t(np.stack([c.viewmat() for c in cams]))
    Ks = _t(np.stack([c.K for c in cams]))
    fr = FrameRenderer(t, W, H, render_mode="RGB+ED", frames_in_flight=3, sizing_camera=(cams[0].viewmat(), cams[0].K),
                       capacity_margin=1.6)
    colors, alphas, mine = render_sharded(t, vms, Ks, W, H, gather=False, renderer=fr, render_mode="RGB+ED")
    assert list(mine) == list(range(8)) and colors.shape == (8, H, W, 4) and alphas.shape == (8, H, W, 1)
    for i in (0, 5):
        ref, ra, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs,
                                           cams[i].viewmat(), cams[i].K, W, H, 3, with_depth=True, flip_eps=O.EPS_PATH)
        ref = ref.astype(np.float64)
        ref[..., 3] /= np.maximum(ra, 1e-10)                  # the port returns the depth sum ("D")
        st = O.check_frame(colors[i].cpu().numpy(), alphas[i].cpu().numpy(), ref, ra, info["margins"], O.EPS_PATH,
                           info["edge_mask"], expected_depth=True, what=f"ring camera {block[i]}",
                           flip_weight=info["flip_weight"], feat_max=info["feat_max"], require_flip_bound=True)
        print(f"\nring camera {block[i]}: {st}")


def test_config3_backward_directional_derivatives(config2):
    """configs[2] at full size (1 M Gaussians, 1920x1080, forward + backward): the analytic
    gradient must predict what the forward does along random parameter directions
    (size-independent property; the HIP path checked against itself in double-precision sums)."""
    from robosimgs_amd import rasterization
    g, cam, t = config2
    vm, K = _t(cam.viewmat())[None], _t(cam.K)[None]
    gen = torch.Generator(DEV).manual_seed(11)
    wimg = torch.randn(1, 1080, 1920, 3, device=DEV, generator=gen)

    def loss_of(p):
        c, a, _ = rasterization(p["means"], p["quats"], p["scales"], p["opacities"], p["colors"], vm, K,
                                1920, 1080, sh_degree=3)
        return c, (c.double() * wimg.double()).sum()

    names = ("means", "quats", "scales", "opacities", "colors")
    p = {k: t[k].clone().requires_grad_(True) for k in names}
    c, _ = loss_of(p)
    (c * wimg).sum().backward()
    grads = {k: p[k].grad.double() for k in names}
    assert all(torch.isfinite(v).all() for v in grads.values())
    # Directions are sign(gradient) x U(0.5, 1): the directional derivative is then a sum of
    # like-signed terms, far above the noise of kinks (clamp_min, alpha thresholds) and of fp32
    # evaluation, so a wrong gradient scale or a missing term shows up directly.
    for k, eps, tol in (("colors", 2e-3, 1e-2), ("opacities", 1e-3, 2e-2), ("scales", 2e-5, 3e-2),
                        ("means", 1e-4, 3e-2), ("quats", 1e-3, 3e-2)):
        u = torch.rand(t[k].shape, device=DEV, generator=gen) * 0.5 + 0.5
        d = torch.sign(grads[k]).float() * u
        q_hi = {n: (t[n] + eps * d if n == k else t[n]) for n in names}
        q_lo = {n: (t[n] - eps * d if n == k else t[n]) for n in names}
        with torch.no_grad():
            fd = (loss_of(q_hi)[1] - loss_of(q_lo)[1]) / (2 * eps)
        an = (grads[k] * d.double()).sum()
        rel = abs(float(fd - an)) / (abs(float(an)) + 1e-12)
        assert rel <= tol, f"{k}: finite difference {float(fd):.6e} vs analytic {float(an):.6e} (rel {rel:.3e})"
