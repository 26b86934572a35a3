"""Device-side similarity transforms of Gaussian groups against the host-side
NumPy `Gaussians.transformed()` and against the renderer itself (equivariance)."""
import math

import numpy as np
import pytest
import torch

from robosimgs_amd import camera_ring, synthetic_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rot(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


@pytest.mark.parametrize("deg,scale", [(3, 1.0), (2, 1.7), (0, 0.5)])
def test_single_transform_matches_host(deg, scale):
    from robosimgs_amd import transform_gaussians
    rng = np.random.default_rng(deg)
    g = synthetic_scene(5000, math.log(0.05), deg, 2)
    R, t = _rot(rng), rng.normal(size=3)
    ref = g.transformed(R, t, scale).to_torch(DEV, deg)
    got = transform_gaussians(g.to_torch(DEV, deg), [R], [t], [scale])
    for k in ("means", "scales", "opacities"):
        torch.testing.assert_close(got[k], ref[k], rtol=2e-5, atol=2e-5)
    # quaternions up to sign / norm conventions: compare the rotation they encode
    a = got["quats"] / got["quats"].norm(dim=1, keepdim=True)
    b = ref["quats"] / ref["quats"].norm(dim=1, keepdim=True)
    assert float((1 - (a * b).sum(1).abs()).max()) < 1e-5
    kc = (deg + 1) ** 2
    torch.testing.assert_close(got["colors"][:, :kc], ref["colors"][:, :kc], rtol=1e-4, atol=2e-5)


def test_groups_static_rows_and_in_place():
    from robosimgs_amd import transform_gaussians
    rng = np.random.default_rng(7)
    n = 7001
    g = synthetic_scene(n, math.log(0.05), 3, 5)
    t0 = g.to_torch(DEV, 3)
    gid = torch.from_numpy(rng.integers(-1, 3, size=n).astype(np.int32)).to(DEV)     # -1 static, 3 groups
    Rs = [_rot(rng) for _ in range(3)]
    ts = [rng.normal(size=3) for _ in range(3)]
    ss = [1.0, 1.3, 0.8]
    got = transform_gaussians(t0, Rs, ts, ss, group_ids=gid)
    for k in range(-1, 3):
        sel = (gid == k)
        if k < 0:
            for name in ("means", "quats", "scales", "colors"):
                assert torch.equal(got[name][sel], t0[name][sel]), name        # untouched bit for bit
            continue
        ref = g.transformed(Rs[k], ts[k], ss[k]).to_torch(DEV, 3)
        torch.testing.assert_close(got["means"][sel], ref["means"][sel], rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(got["scales"][sel], ref["scales"][sel], rtol=2e-5, atol=1e-6)
        torch.testing.assert_close(got["colors"][sel], ref["colors"][sel], rtol=1e-4, atol=2e-5)
    # in place: same result, the static rows are not even written
    t1 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in t0.items()}
    same = transform_gaussians(t1, Rs, ts, ss, group_ids=gid, out=t1)
    for name in ("means", "quats", "scales", "colors"):
        assert same[name].data_ptr() == t1[name].data_ptr()
        assert torch.equal(same[name], got[name]), name
    # pre-packed device transforms (no host work per call) give the same bits
    from robosimgs_amd.transform import pack_transforms
    xp, rp = pack_transforms(Rs, ts, ss, 3)
    pk = transform_gaussians(t0, group_ids=gid, packed=(_t(xp), _t(rp)))
    for name in ("means", "quats", "scales", "colors"):
        assert torch.equal(pk[name], got[name]), name
    # without SH rotation the coefficient tensor is passed through
    plain = transform_gaussians(t0, Rs, ts, ss, group_ids=gid, rotate_sh=False)
    assert plain["colors"] is t0["colors"] and torch.equal(plain["means"], got["means"])


def test_render_is_equivariant_under_a_moved_scene_and_camera():
    """Moving the whole scene AND the camera by the same rigid motion must not change the image
    (this is what rotating the SH colour field with the Gaussians buys)."""
    from robosimgs_amd import rasterization, transform_gaussians
    rng = np.random.default_rng(3)
    g = synthetic_scene(20_000, math.log(0.04), 3, 1)
    cam = camera_ring(1, 320, 200, thetas=[0.9])[0]
    t0 = g.to_torch(DEV, 3)
    R, t = _rot(rng), rng.normal(size=3)
    moved = transform_gaussians(t0, [R], [t])
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    vm0 = cam.viewmat()
    vm1 = vm0 @ np.linalg.inv(T)                      # world-to-camera of the moved camera
    args = dict(sh_degree=3, render_mode="RGB+ED")
    a, aa, _ = rasterization(t0["means"], t0["quats"], t0["scales"], t0["opacities"], t0["colors"],
                             _t(vm0)[None], _t(cam.K)[None], 320, 200, **args)
    b, ba, _ = rasterization(moved["means"], moved["quats"], moved["scales"], moved["opacities"],
                             moved["colors"], _t(vm1)[None], _t(cam.K)[None], 320, 200, **args)
    bad = ((a - b).abs().amax(-1) > 2e-3) | ((aa - ba).abs()[..., 0] > 2e-3)
    assert float(bad.float().mean()) < 2e-3, float(bad.float().mean())


def test_frame_renderer_poses_groups_per_frame():
    """Dynamic scene through FrameRenderer: every submit poses the groups (in-graph
    mgs_transform_gaussians on the slot's own copy); frames equal transform + rasterization done
    by hand, with three frames in flight and a different pose per frame."""
    from robosimgs_amd import FrameRenderer, rasterization, transform_gaussians
    rng = np.random.default_rng(11)
    n = 30_000
    g = synthetic_scene(n, math.log(0.05), 3, 6)
    t0 = g.to_torch(DEV, 3)
    gid = torch.from_numpy(rng.integers(-1, 2, size=n).astype(np.int32)).to(DEV)
    cams = camera_ring(5, 256, 160)
    r = FrameRenderer(t0, 256, 160, render_mode="RGB+ED", frames_in_flight=3, isect_capacity=600_000,
                      group_ids=gid, n_groups=2)
    poses = [([_rot(rng), _rot(rng)], [0.3 * rng.normal(size=3), 0.3 * rng.normal(size=3)]) for _ in cams]
    tickets, got = [], []
    for cam, (Rs, ts) in zip(cams, poses):
        if len(tickets) == 3:
            tk = tickets.pop(0)
            f = r.fetch(tk)
            got.append((f["colors"].clone(), f["alphas"].clone()))
            r.release(tk)
        tickets.append(r.submit(cam.viewmat(), cam.K, rotations=Rs, translations=ts))
    for tk in tickets:
        f = r.fetch(tk)
        got.append((f["colors"].clone(), f["alphas"].clone()))
        r.release(tk)
    for (c, a), cam, (Rs, ts) in zip(got, cams, poses):
        # (the renderer keeps its own Morton-ordered copy of the scene, group ids permuted with it: r.t, r.group_ids)
        posed = transform_gaussians(r.t, Rs, ts, group_ids=r.group_ids)
        rc, ra, _ = rasterization(posed["means"], posed["quats"], posed["scales"], posed["opacities"],
                                  posed["colors"], _t(cam.viewmat())[None], _t(cam.K)[None], 256, 160,
                                  sh_degree=3, render_mode="RGB+ED")
        assert torch.equal(c, rc[0]) and torch.equal(a, ra[0])
    # a static renderer refuses poses
    static = FrameRenderer(t0, 256, 160, frames_in_flight=1, isect_capacity=600_000)
    with pytest.raises(ValueError):
        static.submit(cams[0].viewmat(), cams[0].K, rotations=[np.eye(3)], translations=[[0, 0, 0]])
