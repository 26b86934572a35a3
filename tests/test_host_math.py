"""CPU logic test of the per-Gaussian DEVICE math (robosimgs_amd/csrc/mgs_math.h compiled
with g++) against the oracle: forward vs the NumPy fp64 restatement, backward vs autograd of
the torch restatement.  No GPU, no product fallback: the harness is built into a temp dir."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import gs_oracle_np as O
from oracle import gs_oracle_torch as OT
from robosimgs_amd import camera_ring, synthetic_scene

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hh(tmp_path_factory):
    so = tmp_path_factory.mktemp("hh") / "libhh.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC",
                    os.path.join(HERE, "host_harness", "harness.cpp"), "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _scene(n=2000, mu=0.1, w=160, h=120, theta=0.7):
    g = synthetic_scene(n, math.log(mu), 3, 11)
    cam = camera_ring(1, w, h, thetas=[theta], radius=5.0)[0]
    return g, cam, w, h


def _run_project(hh, g, cam, w, h):
    n = len(g)
    radii = np.zeros(n, np.int32)
    m2d, dep, con, comp = (np.zeros((n, 2), np.float32), np.zeros(n, np.float32),
                           np.zeros((n, 3), np.float32), np.zeros(n, np.float32))
    vm, K = _f(cam.viewmat()), _f(cam.K)
    hh.hh_project(n, _p(_f(g.means)), _p(_f(g.quats)), _p(_f(g.scales)), _p(vm), _p(K), w, h,
                  ctypes.c_float(0.3), ctypes.c_float(0.01), ctypes.c_float(1e10),
                  ctypes.c_float(0.0), _p(radii), _p(m2d), _p(dep), _p(con), _p(comp))
    return radii, m2d, dep, con, comp


@pytest.mark.parametrize("antialiased,with_opacity", [(False, True), (True, True), (False, False)])
def test_projection_opacity_aware_radius_rule_matches_oracle(hh, antialiased, with_opacity):
    """the gsplat >= 1.5 radius rule as the device math compiles it -- per-axis extents
    min(3.33, sqrt(2 ln(255 o))) sqrt(Sigma_ii), cull below 1/255 -- against the NumPy restatement."""
    g, cam, w, h = _scene(n=4000)
    n = len(g)
    op = _f(g.opacities)
    op[::7] = 0.003                      # below 1/255: culled by the rule
    op[1::7] = 0.0045                    # barely above: tiny extents
    rx, ry, m2d = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, 2), np.float32)
    vm, K = _f(cam.viewmat()), _f(cam.K)
    hh.hh_project_rule(n, _p(_f(g.means)), _p(_f(g.quats)), _p(_f(g.scales)), _p(vm), _p(K), w, h,
                       ctypes.c_float(0.3), ctypes.c_float(0.01), ctypes.c_float(1e10), ctypes.c_float(0.0),
                       _p(op) if with_opacity else None, int(antialiased), _p(rx), _p(ry), _p(m2d))
    ref = O.project(g.means, g.quats, g.scales, cam.viewmat(), cam.K, w, h, radius_rule="opacity_aware",
                    opacities=op.astype(np.float64) if with_opacity else None, antialiased=antialiased)
    rr = ref["radii"]
    assert rr.shape == (n, 2) and (rr[:, 0] > 0).sum() > 500
    assert ((rx > 0) == (ry > 0)).all()                              # both extents or neither
    assert ((rx > 0) != (rr[:, 0] > 0)).sum() <= 1
    both = (rx > 0) & (rr[:, 0] > 0)
    assert (np.abs(rx[both] - rr[both, 0]) > 0).sum() + (np.abs(ry[both] - rr[both, 1]) > 0).sum() <= 3   # ceil knife edges
    assert np.abs(np.stack([rx, ry], -1)[both] - rr[both]).max() <= 1
    if with_opacity:
        assert not (rx[::7] > 0).any()                              # opacity < 1/255
        cl = O.project(g.means, g.quats, g.scales, cam.viewmat(), cam.K, w, h)
        vis = both & (cl["radii"] > 0)
        # the opacity-aware box never exceeds 3.33 / 3 of the classic radius (sqrt(Sigma_ii) <= sqrt(lambda_1))
        assert (np.maximum(rx, ry)[vis] <= np.ceil(cl["radii"][vis] * (3.33 / 3.0)) + 1).all()
        assert np.maximum(rx, ry)[vis].astype(np.int64).sum() < 0.9 * cl["radii"][vis].astype(np.int64).sum()


def test_projection_forward_matches_oracle(hh):
    g, cam, w, h = _scene()
    radii, m2d, dep, con, comp = _run_project(hh, g, cam, w, h)
    ref = O.project(g.means, g.quats, g.scales, cam.viewmat(), cam.K, w, h)
    assert (radii > 0).sum() > 500
    assert ((radii > 0) != (ref["radii"] > 0)).sum() <= 1
    both = (radii > 0) & (ref["radii"] > 0)
    assert (np.abs(radii[both] - ref["radii"][both]) > 0).sum() <= 2
    np.testing.assert_allclose(m2d[both], ref["means2d"][both], rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(con[both], ref["conics"][both], rtol=3e-4, atol=1e-6)
    np.testing.assert_allclose(comp[both], ref["compensations"][both], rtol=3e-4, atol=1e-6)


def test_projection_from_the_2x3_factor_is_the_same_function_and_tighter_on_needles(tmp_path):
    """csrc/mgs_math.h -DMGS_PROJ_FACTORED=1 (a build knob, not the default: profiles/r6/00_experiments.md section 9): cov2d from
    J R Rq S instead of the textbook order -- the same radii, means and compensation, and on needle-like Gaussians a conic whose
    error does not grow with the axis ratio."""
    from robosimgs_amd import synthetic_scene_heavy_tailed
    libs = {}
    for name, flag in (("textbook", "-DMGS_PROJ_FACTORED=0"), ("factored", "-DMGS_PROJ_FACTORED=1")):
        so = tmp_path / f"libhh_{name}.so"
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", flag,
                        os.path.join(HERE, "host_harness", "harness.cpp"), "-o", str(so)], check=True)
        libs[name] = ctypes.CDLL(str(so))
    g = synthetic_scene_heavy_tailed(40_000, sh_degree=0, seed=2)
    w, h = 640, 360
    cam = camera_ring(1, w, h, thetas=[0.3])[0]
    ref = O.project(g.means.astype(np.float64), g.quats.astype(np.float64), g.scales.astype(np.float64),
                    _f(cam.viewmat()).astype(np.float64), _f(cam.K).astype(np.float64), w, h)
    out = {k: _run_project(L, g, cam, w, h) for k, L in libs.items()}
    rt, rf = out["textbook"][0], out["factored"][0]
    both = (rt > 0) & (rf > 0) & (ref["radii"] > 0)
    assert both.sum() > 10_000 and ((rt > 0) != (rf > 0)).sum() <= 2 and (rt[both] != rf[both]).sum() <= 2
    np.testing.assert_allclose(out["factored"][1][both], out["textbook"][1][both], rtol=0, atol=0)        # means2d: untouched code
    np.testing.assert_allclose(out["factored"][4][both], ref["compensations"][both], rtol=2e-5, atol=1e-6)   # (the textbook order's det0 cancels too)
    err = {k: (np.abs(v[3].astype(np.float64) - ref["conics"]).max(1) / np.abs(ref["conics"]).max(1).clip(1e-30))[both] for k, v in out.items()}
    assert err["factored"].max() < 1e-5, err["factored"].max()
    assert err["textbook"].max() > 20 * err["factored"].max()        # (the scene does hold needles: the textbook order shows them)
    assert abs(np.median(err["factored"]) / np.median(err["textbook"]) - 1) < 0.5


def test_projection_backward_matches_autograd(hh):
    g, cam, w, h = _scene(1500)
    n = len(g)
    radii, m2d, dep, con, comp = _run_project(hh, g, cam, w, h)
    rng = np.random.default_rng(0)
    v_m2d, v_dep = rng.normal(size=(n, 2)), rng.normal(size=n)
    v_con, v_comp = rng.normal(size=(n, 3)), rng.normal(size=n)
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)
    means, quats, scales, vm = t(g.means), t(g.quats), t(g.scales), t(cam.viewmat())
    p = OT.project(means, quats, scales, vm, torch.tensor(cam.K), w, h)
    vis = (p["radii"].numpy() > 0) & (radii > 0)
    mask = torch.tensor(vis.astype(np.float64))
    loss = ((p["means2d"] * torch.tensor(v_m2d)).sum(-1) * mask).sum() \
        + (p["depths"] * torch.tensor(v_dep) * mask).sum() \
        + ((p["conics"] * torch.tensor(v_con)).sum(-1) * mask).sum() \
        + (p["compensations"] * torch.tensor(v_comp) * mask).sum()
    loss.backward()
    r_eff = np.where(vis, radii, 0).astype(np.int32)
    v_means, v_quats, v_scales = (np.zeros((n, 3), np.float32), np.zeros((n, 4), np.float32),
                                  np.zeros((n, 3), np.float32))
    v_R, v_t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    hh.hh_project_vjp(n, _p(_f(g.means)), _p(_f(g.quats)), _p(_f(g.scales)), _p(_f(cam.viewmat())),
                      _p(_f(cam.K)), w, h, ctypes.c_float(0.3), _p(r_eff), _p(con), _p(comp),
                      _p(_f(v_m2d)), _p(_f(v_dep)), _p(_f(v_con)), _p(_f(v_comp)), _p(v_means),
                      _p(v_quats), _p(v_scales), _p(v_R), _p(v_t))

    def close(a, b, name):
        scale = np.abs(b).max(axis=-1, keepdims=True) + 1e-3 * np.abs(b).max() + 1e-12
        err = (np.abs(a - b) / scale).max()
        assert err < 2e-3, f"{name}: max scaled error {err:.3e}"
    close(v_means[vis], means.grad.numpy()[vis], "v_means")
    close(v_quats[vis], quats.grad.numpy()[vis], "v_quats")
    close(v_scales[vis], scales.grad.numpy()[vis], "v_scales")
    gv = vm.grad.numpy()
    close(v_R.reshape(1, 9), gv[:3, :3].reshape(1, 9), "v_viewmat R")
    close(v_t.reshape(1, 3), gv[:3, 3].reshape(1, 3), "v_viewmat t")


@pytest.mark.parametrize("deg,K", [(0, 1), (1, 4), (2, 9), (3, 16), (2, 16)])
def test_sh_forward_backward(hh, deg, K):
    rng = np.random.default_rng(deg)
    n = 300
    dirs, coeffs, v_rgb = rng.normal(size=(n, 3)), rng.normal(size=(n, K, 3)), rng.normal(size=(n, 3))
    colors, v_coeffs, v_dirs = (np.zeros((n, 3), np.float32), np.zeros((n, K, 3), np.float32),
                                np.zeros((n, 3), np.float32))
    hh.hh_sh(n, deg, K, _p(_f(dirs)), _p(_f(coeffs)), _p(_f(v_rgb)), _p(colors), _p(v_coeffs), _p(v_dirs))
    d, c = torch.tensor(dirs, requires_grad=True), torch.tensor(coeffs, requires_grad=True)
    out = OT.spherical_harmonics(deg, d, c)
    (out * torch.tensor(v_rgb)).sum().backward()
    np.testing.assert_allclose(colors, out.detach().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(v_coeffs, c.grad.numpy(), rtol=1e-4, atol=2e-5)
    d_ref = d.grad.numpy() if d.grad is not None else np.zeros_like(dirs)   # degree 0: constant
    np.testing.assert_allclose(v_dirs, d_ref, rtol=2e-3, atol=2e-4)


def test_sh_basis_independent_formula():
    """The oracle's basis vs the textbook real SH in spherical angles (degree <= 2)."""
    rng = np.random.default_rng(3)
    d = rng.normal(size=(200, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x, y, z = d.T
    Y = O.sh_basis(2, d)
    c = math.sqrt
    ref = np.stack([np.full_like(x, 0.5 * c(1 / math.pi)),
                    -c(3 / (4 * math.pi)) * y, c(3 / (4 * math.pi)) * z, -c(3 / (4 * math.pi)) * x,
                    0.5 * c(15 / math.pi) * x * y, -0.5 * c(15 / math.pi) * y * z,
                    0.25 * c(5 / math.pi) * (3 * z * z - 1), -0.5 * c(15 / math.pi) * x * z,
                    0.25 * c(15 / math.pi) * (x * x - y * y)], axis=1)
    np.testing.assert_allclose(Y, ref, atol=1e-12)
