"""oracle/gs_cpu.cpp (the fp32 C++/OpenMP port that serves as bench.py's cpu_baseline) against
the fp64 NumPy oracle and the stored golden; BASELINE.json configs[0] is its plumbing case."""
import math
import os

import numpy as np

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
