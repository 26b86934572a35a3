"""oracle/gs_cpu.cpp (the fp32 C++/OpenMP port that serves as bench.py's cpu_baseline) against
the fp64 NumPy oracle and the stored golden; BASELINE.json configs[0] is its plumbing case."""
import math
import os

import numpy as np
import pytest

from oracle import cpu_ref
from oracle import gs_oracle_np as O
from robosimgs_amd import camera_ring, synthetic_scene

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "render_small.npz"))


def test_config0_counts_and_image():
    """10k random Gaussians, SH degree 0, 256x256 (BASELINE configs[0]): the work counters the
    survey calibrated (n_vis 9,849; n_isect 37,024) and the image."""
    g = synthetic_scene(10_000, math.log(0.05), 0, 0)
    cam = camera_ring(1, 256, 256, thetas=[0.3])[0]
    img, alpha, info = cpu_ref.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs,
                                      cam.viewmat(), cam.K, 256, 256, 0, n_threads=4)
    vm, K = cam.viewmat().astype(np.float32).astype(np.float64), cam.K.astype(np.float32).astype(np.float64)
    ref, ref_alpha, meta = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs,
                                    vm, K, 256, 256, sh_degree=0, margins=True, flip_eps=O.EPS_PATH)
    assert info["n_vis"] == meta["n_vis"] == 9849
    assert info["n_isect"] == meta["n_isect"] == 37024
    assert info["pair_evals"] == meta["pair_evals"]
    # fp32 port vs fp64 oracle: zero pixels over 1e-4 that no threshold within EPS_PATH explains
    st = O.check_frame(img, alpha, ref, ref_alpha, meta["margins"], O.EPS_PATH, meta["edge_mask"], what="fp32 port",
                       flip_weight=meta["flip_weight"], feat_max=meta["feat_max"], require_flip_bound=True)
    assert st["flip_over_bound"] == 0
    # fp64 instantiation of the port == the NumPy oracle (to the fp32 rounding of its outputs),
    # image and margins alike
    r64, a64, i64 = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vm, K, 256, 256, 0,
                                       n_threads=4, flip_eps=O.EPS_PATH, want_touched=True)
    # Gaussians blended into a could-flip pixel: some, not all, and only visible ones
    assert 0 < i64["touched"].sum() < 0.5 * meta["n_vis"] and not (i64["touched"] & (meta["radii"] <= 0)).any()
    np.testing.assert_allclose(r64, ref, atol=3e-7)
    np.testing.assert_allclose(a64, ref_alpha[..., 0], atol=3e-7)
    both = np.isfinite(meta["margins"]) & np.isfinite(i64["margins"])
    assert (np.isfinite(meta["margins"]) == np.isfinite(i64["margins"])).all()
    np.testing.assert_allclose(i64["margins"][both], meta["margins"][both], rtol=1e-3, atol=1e-3)
    assert i64["n_edge_gaussians"] == meta["n_edge_gaussians"]
    assert (i64["edge_mask"] == meta["edge_mask"]).all()
    # the two oracles price the near-flip decisions alike
    np.testing.assert_allclose(i64["flip_weight"], meta["flip_weight"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(i64["feat_max"], meta["feat_max"], rtol=1e-6)
    assert 0 < (meta["flip_weight"] > 0).mean() < 0.03


def test_port_opacity_aware_radius_rule_equals_numpy_oracle():
    """SURVEY.md A.4 in the port (gs_cpu_render_f64(radius_rule=1)) == the NumPy restatement of the same rule: counts,
    image, margins, knife-edge classification (per-axis extents, opacities within 1e-5 of 1/255) and flip weights -- what
    lets the full-size tests and the gradient budgets run under the rule as they do under the classic one."""
    g = synthetic_scene(10_000, math.log(0.05), 2, 0)
    g.opacity_logits[::13] = -5.8                      # under 1/255: culled by the rule
    g.opacity_logits[1::13] = -5.45
    cam = camera_ring(1, 240, 176, thetas=[0.3])[0]
    vm, K = cam.viewmat().astype(np.float32).astype(np.float64), cam.K.astype(np.float32).astype(np.float64)
    ref, ref_alpha, meta = O.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vm, K, 240, 176, sh_degree=2,
                                    render_mode="RGB+D", margins=True, flip_eps=O.EPS_PATH, radius_rule="opacity_aware")
    r64, a64, i64 = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vm, K, 240, 176, 2,
                                       with_depth=True, n_threads=4, flip_eps=O.EPS_PATH, want_projected=True,
                                       radius_rule="opacity_aware")
    assert i64["n_vis"] == meta["n_vis"] and i64["n_isect"] == meta["n_isect"]
    assert ((i64["radii"] > 0) == (meta["radii"][:, 0] > 0)).all() and (i64["radii"] == meta["radii"][:, 0]).all()
    assert not (i64["radii"][::13] > 0).any()
    # two fp64 implementations agree to the fp32 rounding of the port's outputs, except where the blend's alpha test sits
    # on its threshold to 1e-6 relative (this scene holds one such pixel: |255 alpha - 1| = 3e-8)
    bad = (np.abs(r64 - ref).max(axis=-1) > 3e-6) | (np.abs(a64 - ref_alpha[..., 0]) > 3e-7)
    assert bad.sum() <= 2 and (meta["margins"][:3][:, bad].min(axis=0) < 1e-6).all()
    assert (np.isfinite(meta["margins"]) == np.isfinite(i64["margins"])).all()
    assert i64["n_edge_gaussians"] == meta["n_edge_gaussians"]
    assert (i64["edge_mask"] == meta["edge_mask"]).all()
    np.testing.assert_allclose(i64["flip_weight"][~bad], meta["flip_weight"][~bad], rtol=2e-3, atol=1e-6)
    # and it is another frame than the classic rule's: fewer pairs, a different image
    _, _, c64 = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vm, K, 240, 176, 2,
                                   with_depth=True, n_threads=4, margins=False)
    assert i64["n_isect"] < 0.8 * c64["n_isect"]


def test_port_backward_matches_autograd_oracle():
    """A.2 step 10 in the C++ port (fp64 sums) == autograd of oracle/gs_oracle_torch.py's blend on the
    port's own projected quantities, RGB+D with a background."""
    import torch
    from oracle import gs_oracle_torch as OT
    g = GOLD
    W, H, deg = int(g["width"]), int(g["height"]), int(g["sh_degree"])
    rng = np.random.default_rng(8)
    w_img = rng.normal(size=(H, W, 4)).astype(np.float32)
    w_a = rng.normal(size=(H, W)).astype(np.float32)
    bg = np.array([0.2, 0.4, 0.6, 0.1], np.float32)
    _, _, info = cpu_ref.render_f64(g["means"], g["quats"], g["scales"], g["opacities"], g["sh_coeffs"],
                                    g["viewmat"], g["K"], W, H, deg, with_depth=True, background=bg,
                                    v_render=w_img, v_alpha=w_a, want_projected=True, n_threads=2)
    t = lambda x: torch.tensor(np.asarray(x, dtype=np.float64), requires_grad=True)
    m2, con, ft, op = t(info["means2d"]), t(info["conics"]), t(info["feats"]), t(g["opacities"].astype(np.float32))
    tw, th = -(-W // 16), -(-H // 16)
    _, iid, fid = O.isect_tiles(info["means2d"], info["radii"], info["feats"][:, 3], 16, tw, th)
    offs = O.isect_offsets(iid, 1, tw, th)[0]
    img, al = OT.rasterize(m2, con, ft, op, fid, offs, W, H, 16, torch.tensor(bg.astype(np.float64)))
    ((img * torch.tensor(w_img.astype(np.float64))).sum() + (al * torch.tensor(w_a.astype(np.float64))).sum()).backward()
    for name, a, b in (("means2d", m2.grad, info["g_means2d"]), ("conics", con.grad, info["g_conics"]),
                       ("feats", ft.grad, info["g_feats"]), ("opacities", op.grad, info["g_opacities"])):
        a = a.numpy()
        assert np.abs(a - b).max() <= 1e-10 * (1 + np.abs(a).max()), name


def test_golden_scene_with_depth_and_background():
    g = GOLD
    W, H, deg = int(g["width"]), int(g["height"]), int(g["sh_degree"])
    img, alpha, info = cpu_ref.render(g["means"], g["quats"], g["scales"], g["opacities"],
                                      g["sh_coeffs"], g["viewmat"], g["K"], W, H, deg,
                                      with_depth=True, background=g["background"], n_threads=2)
    assert info["n_isect"] == int(g["n_isect"]) and info["n_vis"] == int(g["n_vis"])
    ed = g["RGB_ED_bg_image"].copy()
    a = g["RGB_ED_alpha"][..., 0]
    ed[..., 3] *= np.maximum(a, 1e-10)                     # port returns accumulated depth ("D")
    np.testing.assert_allclose(img, ed, atol=2e-4)
    np.testing.assert_allclose(alpha, a, atol=1e-4)


def test_thread_count_does_not_change_the_image():
    g = synthetic_scene(3000, math.log(0.1), 1, 4)
    cam = camera_ring(1, 96, 64, thetas=[2.0])[0]
    args = (g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, cam.viewmat(), cam.K, 96, 64, 1)
    a, _, _ = cpu_ref.render(*args, n_threads=1)
    b, _, _ = cpu_ref.render(*args, n_threads=8)
    np.testing.assert_array_equal(a, b)


def _blend_grads(g, cam, W, H, deg, w_img, w_a, thresholds=None, want_budget=False):
    vm, K = cam.viewmat().astype(np.float32), cam.K.astype(np.float32)
    img, al, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs,
                                       vm, K, W, H, deg, with_depth=True, v_render=w_img, v_alpha=w_a,
                                       flip_eps=O.EPS_PATH_GRAD, want_touched=True, want_budget=want_budget, n_threads=4,
                                       thresholds=thresholds)
    info["image"], info["alpha"] = img, al
    return info


def test_gradient_budget_covers_real_flips_and_rejects_rows_the_old_gate_let_through():
    """tests/grad_gate.py.  (1) REAL flips: the same blend with its thresholds moved by less than the gate's eps
    (alpha >= (1 +- 2.5e-4) / 255, stop at T' <= (1 +- 2e-3) 1e-4) flips exactly the decisions the margins call "could
    flip" and nothing else; every gradient row must then lie within rounding + 1.5 x the row's flip budget of the
    unperturbed one, and every pixel within the forward's flip-weight bound -- both bounds hold against what flips
    really do.  (2) A backward that is WRONG on Gaussians near a threshold (here: every row of ~100
    touched Gaussians doubled) passed the round-3 gate, which exempted touched rows altogether, and fails this one."""
    import pytest
    from grad_gate import compare
    W, H, deg = 160, 112, 1
    g = synthetic_scene(12000, math.log(0.07), deg, 9)
    cam = camera_ring(1, W, H, thetas=[0.9])[0]
    rng = np.random.default_rng(3)
    w_img = rng.normal(size=(H, W, 4)).astype(np.float32)
    w_a = rng.normal(size=(H, W)).astype(np.float32)
    ref = _blend_grads(g, cam, W, H, deg, w_img, w_a, want_budget=True)
    bud, touched = ref["budget"], ref["touched"]
    vis = ref["radii"] > 0
    assert bud.shape == (len(g), 4) and (bud >= 0).all() and not bud[~touched].any()
    assert 0.02 < touched[vis].mean() < 0.9 and (bud[touched] > 0).any(axis=1).mean() > 0.9
    rows = {"means2d": (ref["g_means2d"], bud[:, 0]), "conics": (ref["g_conics"], bud[:, 1]),
            "feats": (ref["g_feats"], bud[:, 2]), "opacities": (ref["g_opacities"].reshape(-1, 1), bud[:, 3])}
    # (1) real flips
    moved = 0
    eps = O.EPS_PATH_GRAD
    for thr in ((1 + 0.8 * eps["alpha"], 1.0), (1 - 0.8 * eps["alpha"], 1.0), (1.0, 1 + 0.6 * eps["T"]), (1.0, 1 - 0.6 * eps["T"]),
                (1 + 0.8 * eps["alpha"], 1 - 0.6 * eps["T"])):
        per = _blend_grads(g, cam, W, H, deg, w_img, w_a, thresholds=thr)
        # the image moved at could-flip pixels only, by no more than the flip weight allows
        d_img = np.abs(per["image"].astype(np.float64) - ref["image"])
        fmax = np.abs(ref["feats"][vis]).max(axis=0)
        assert (d_img <= 1e-6 + 1.5 * ref["flip_weight"][..., None] * 2 * fmax).all()
        assert (np.abs(per["alpha"].astype(np.float64) - ref["alpha"]) <= 1e-6 + 1.5 * ref["flip_weight"]).all()
        for name, (r, b) in rows.items():
            key = "g_" + name if name != "opacities" else "g_opacities"
            st = compare(name + f" (thresholds x {thr})", per[key].reshape(r.shape), r, row_tol=1e-7, bad_frac=1.0, cos_min=0.99,
                         budget=b, verbose=False)
            moved += st["rows_over_rounding"]
    assert moved > 0, "the perturbation flipped nothing: the case does not exercise the budget"
    # (2) a backward wrong on touched rows only
    cand = np.flatnonzero(touched & vis)
    idx = cand[::max(1, int(math.ceil(len(cand) / (0.008 * len(g)))))]        # under the old gate's 1 % allowance
    assert 50 < len(idx) <= 0.01 * len(g)
    for name, (r, b) in rows.items():
        bad = r.copy()
        bad[idx] *= 2.0
        compare(name + " (old gate)", bad, r, row_tol=2e-3, bad_frac=1e-2, cos_min=0.9, touched=touched, verbose=False)
        with pytest.raises(AssertionError, match="flip budget"):
            compare(name + " (new gate)", bad, r, row_tol=2e-3, bad_frac=1e-2, cos_min=0.9, budget=b, verbose=False)


def test_heavy_tailed_scene_generator_and_the_fp32_restatement_gate():
    """synthetic_scene_heavy_tailed (NOT a BASELINE config): deterministic, shaped as documented, and -- at a size the port
    finishes in seconds -- ill-conditioned for fp32 the way the full-size GPU test relies on: the port's float
    instantiation is pixels away from its fp64 one where check_frame finds no near-flip decision, and
    check_frame_against_fp32_port accepts a frame that is as close as that restatement and rejects one that is farther."""
    import math
    from robosimgs_amd import synthetic_scene_heavy_tailed, camera_ring
    n = 60_000
    g = synthetic_scene_heavy_tailed(n, math.log(0.02), 1, 3, n_needles=1500, n_screen_filling=3)
    g2 = synthetic_scene_heavy_tailed(n, math.log(0.02), 1, 3, n_needles=1500, n_screen_filling=3)
    assert np.array_equal(g.means, g2.means) and np.array_equal(g.log_scales, g2.log_scales) and len(g) == n
    s = np.sort(g.scales, axis=1)
    assert (s[-1503:-3, 2] / s[-1503:-3, 0]).min() >= 99.0                       # the needles: 100-400 : 1
    assert g.scales[-3:].min() >= 1.5 and g.opacities[-3:].max() <= 0.2          # the screen-filling ones: large and faint
    assert np.std(g.log_scales[:-1503].mean(axis=1)) > 1.0                       # heavy-tailed extents (0.4 / sqrt 3)
    with pytest.raises(ValueError):
        synthetic_scene_heavy_tailed(100)
    W, H = 480, 272
    cam = camera_ring(1, W, H, thetas=[0.3])[0]
    vm32, K32 = np.asarray(cam.viewmat(), np.float32), np.asarray(cam.K, np.float32)
    ref, ra, info = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vm32, K32, W, H, 1, with_depth=True,
                                       flip_eps=O.EPS_PATH)
    r32, a32, i32 = cpu_ref.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vm32, K32, W, H, 1, with_depth=True)
    assert i32["n_vis"] > 0.3 * n and info["n_isect"] > 2 * i32["n_vis"]
    st = O.check_frame_against_fp32_port(r32, a32, ref, ra, r32, a32, info["margins"], O.EPS_PATH, info["edge_mask"], what="fp32 port vs itself")
    assert st["over_tol"] == st["over_tol_fp32_port"]
    print(f"\nheavy-tailed scene at {n} Gaussians, {W}x{H}: fp32 port vs fp64 port {st}")
    worse = r32.copy()
    worse[::7, ::5, 0] += 3e-4                                                    # a frame that is farther off than the restatement
    with pytest.raises(AssertionError):
        O.check_frame_against_fp32_port(worse, a32, ref, ra, r32, a32, info["margins"], O.EPS_PATH, info["edge_mask"], what="corrupted")


def test_counting_sort_does_not_depend_on_the_team_the_runtime_starts():
    """The port's parallel counting sort cuts the keys into omp_get_max_threads() slices; when the runtime starts fewer
    threads than that (OMP_THREAD_LIMIT) every slice must still be counted and scattered: same frame, same counters."""
    import subprocess
    import sys
    code = ("import math, numpy as np\n"
            "from oracle import cpu_ref\n"
            "from robosimgs_amd import camera_ring, synthetic_scene\n"
            "g = synthetic_scene(20_000, math.log(0.05), 3, 0)\n"
            "cam = camera_ring(1, 256, 256, thetas=[0.3])[0]\n"
            "r, a, info = cpu_ref.render(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, cam.viewmat(), cam.K, 256, 256, 3)\n"
            "print(repr(float(np.abs(r).sum())), repr(float(a.sum())), info['n_isect'], info['pair_evals'])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for limit in (None, "3"):
        env = dict(os.environ, OMP_NUM_THREADS="8", PYTHONPATH=root)
        env.pop("OMP_THREAD_LIMIT", None)
        if limit:
            env["OMP_THREAD_LIMIT"] = limit
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, cwd=root, check=True, capture_output=True,
                                   text=True).stdout.strip())
    assert outs[0] == outs[1] and outs[0]


def test_stage_isolated_blend_equals_the_numpy_blend_on_the_same_inputs():
    """cpu_ref.blend_f64 (A.2 steps 9-10 in fp64 on GIVEN fp32 projected quantities and lists) against
    gs_oracle_np.rasterize on the very same arrays: image, alpha, margins, flip weights; and the blend's backward against
    the whole-path entry fed the same scene (there the inputs are fp64: gradients agree to the inputs' fp32 rounding)."""
    g = synthetic_scene(3_000, math.log(0.08), 1, 3)
    W, H = 112, 80
    cam = camera_ring(1, W, H, thetas=[1.1])[0]
    vm, K = cam.viewmat().astype(np.float32).astype(np.float64), cam.K.astype(np.float32).astype(np.float64)
    p = O.project(g.means, g.quats, g.scales, vm, K, W, H)
    rgb = O.sh_colors(1, g.means, O.campos_from_viewmat(vm), g.sh_coeffs)
    feats = np.concatenate([rgb, p["depths"][:, None]], axis=1)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    m2d, con, opa, fe, dep = f32(p["means2d"]), f32(p["conics"]), f32(g.opacities), f32(feats), f32(p["depths"])
    tw, th = -(-W // 16), -(-H // 16)
    tpg, isect_ids, flat = O.isect_tiles(m2d, p["radii"], dep, 16, tw, th, dtype=np.float32)
    offs = O.isect_offsets(isect_ids, 1, tw, th)[0]
    img, alpha, last, st = O.rasterize(m2d.astype(np.float64), con.astype(np.float64), fe.astype(np.float64), opa.astype(np.float64),
                                       flat, offs, W, H, margins=True, depths=dep.astype(np.float64), flip_eps=O.EPS_STAGE)
    offsets = np.concatenate([offs.reshape(-1), [len(flat)]]).astype(np.int32)
    rng = np.random.default_rng(5)
    v_render, v_alpha = rng.normal(size=(H, W, 4)).astype(np.float32), rng.normal(size=(H, W)).astype(np.float32)
    out, a, info = cpu_ref.blend_f64(m2d, con, opa, fe, flat, offsets, W, H, depths=dep, flip_eps=O.EPS_STAGE, n_threads=4,
                                     v_render=v_render, v_alpha=v_alpha, want_budget=True)
    assert info["n_isect"] == len(flat) and not info["edge_mask"].any()
    np.testing.assert_allclose(out, img, atol=3e-7)
    np.testing.assert_allclose(a, alpha, atol=3e-7)
    fin = np.isfinite(st["margins"])
    assert np.array_equal(fin, np.isfinite(info["margins"]))
    np.testing.assert_allclose(info["margins"][fin], st["margins"][fin], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(info["flip_weight"], st["flip_weight"], atol=2e-6)
    # backward: the whole-path fp64 entry on the same scene blends fp64 inputs; on the fp32-rounded ones the gradients
    # agree to that rounding wherever no decision sits on a threshold
    r64, a64, i64 = cpu_ref.render_f64(g.means, g.quats, g.scales, g.opacities, g.sh_coeffs, vm, K, W, H, 1, with_depth=True,
                                       n_threads=4, v_render=v_render, v_alpha=v_alpha, flip_eps=O.EPS_PATH, want_budget=True)
    for k in ("g_means2d", "g_conics", "g_feats", "g_opacities"):
        ref, got = i64[k], info[k]
        scale = np.abs(ref).max() + 1e-12
        close = np.abs(got - ref) <= 2e-4 * scale + 1e-3 * np.abs(ref)
        assert close.mean() > 0.995, (k, close.mean())
    assert info["budget"].shape == (3_000, 4) and np.isfinite(info["budget"]).all() and info["touched"].any()
