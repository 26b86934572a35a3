"""robosimgs_amd.points (HIP) against oracle/points_np.py."""
import numpy as np
import pytest
import torch

from oracle import points_np as P

pytestmark = pytest.mark.gpu


def _camera(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    c2w = np.eye(4)
    c2w[:3, :3], c2w[:3, 3] = q, rng.normal(size=3)
    return c2w


def test_project_and_unproject_match_oracle():
    from robosimgs_amd import points
    rng = np.random.default_rng(1)
    K = np.array([[400.0, 0, 160], [0, 410.0, 120], [0, 0, 1]])
    c2w = _camera(rng)
    cam_pts = np.column_stack([rng.uniform(-1, 1, 5000), rng.uniform(-1, 1, 5000), rng.uniform(0.5, 6, 5000)])
    pts = P.unproject_pcd(cam_pts, c2w)
    uv, cam, depth = points.project_pcd(pts.astype(np.float32), K, c2w)
    ruv, rcam, rdepth = P.project_pcd(pts, K, c2w)
    assert uv.dtype == np.float32 and uv.shape == (5000, 3) and depth.shape == (5000, 1)
    np.testing.assert_allclose(cam, rcam, atol=2e-5)
    np.testing.assert_allclose(uv, ruv, atol=5e-3)                 # pixels; fp32 vs the fp64 oracle
    np.testing.assert_allclose(depth, rdepth, atol=2e-5)
    np.testing.assert_allclose(points.unproject_pcd(cam, c2w), pts, atol=2e-5)
    # torch in -> torch out, on the device
    tuv, _, _ = points.project_pcd(torch.from_numpy(pts).float().cuda(), K, c2w)
    assert torch.is_tensor(tuv) and tuv.is_cuda and torch.equal(tuv.cpu(), torch.from_numpy(uv))


@pytest.mark.parametrize("h,w,scale,n", [(64, 96, 2, 20000), (48, 50, 3, 4000), (30, 40, 1, 3000), (16, 16, 2, 0)])
def test_depth_map_matches_oracle_bit_for_bit(h, w, scale, n):
    from robosimgs_amd import points
    rng = np.random.default_rng(h * w + n)
    uv = np.column_stack([rng.uniform(-10, w + 10, n), rng.uniform(-10, h + 10, n), np.ones(n)]).astype(np.float32)
    uv[: n // 10] = np.floor(uv[: n // 10]) + 0.5 * (scale == 1)           # exact .5 ties where they survive fp32
    depth = rng.uniform(0.3, 20.0, n).astype(np.float32)
    depth[n // 2:] = np.round(depth[n // 2:], 1)                           # many equal depths: first index wins
    if n:
        depth[:5] = [-1.0, np.nan, 1e10, 2e10, 0.0]
    dm, idx = points.get_depth_map(uv, depth, h, w, bg_depth=1e10, scale=scale)
    rdm, ridx = P.get_depth_map(uv, depth, h, w, bg_depth=1e10, scale=scale)
    assert dm.shape == (h, w) and dm.dtype == np.float32 and idx.dtype == np.int64
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(dm, rdm)


def test_mask_lookup_matches_oracle():
    from robosimgs_amd import points
    rng = np.random.default_rng(5)
    h, w, n = 60, 80, 30000
    mask = (rng.uniform(size=(h, w)) > 0.5).astype(np.float32)
    depth = rng.uniform(1, 3, size=(h, w)).astype(np.float32)
    uv = np.column_stack([rng.uniform(-5, w + 5, n), rng.uniform(-5, h + 5, n)]).astype(np.float32)
    pd = rng.uniform(1, 3, size=(n, 1)).astype(np.float32)
    for args in ((), (depth, pd, 0.4)):
        got = points.mask_pcd_2d(uv, mask, 0.5, *args)
        ref = P.mask_pcd_2d(uv, mask, 0.5, *args)
        assert got.dtype == bool and got.shape == (n,)
        # the sampled value sits exactly on a threshold only by rounding: allow a handful of flips
        assert (got != ref).sum() <= n * 1e-3, (got != ref).sum()


def test_zbuffer_visibility_round_trip():
    """The use the reference's helpers are written for: project a cloud, z-buffer it, and keep
    the points that are the visible surface -- a wall in front of another wall hides it."""
    from robosimgs_amd import points
    h, w = 96, 128
    K = np.array([[100.0, 0, 64], [0, 100.0, 48], [0, 0, 1]])
    c2w = np.eye(4)
    gx, gy = np.meshgrid(np.linspace(-0.6, 0.6, 200), np.linspace(-0.45, 0.45, 150))
    near = np.column_stack([gx.ravel(), gy.ravel(), np.full(gx.size, 1.0)])
    far = np.column_stack([gx.ravel() * 2, gy.ravel() * 2, np.full(gx.size, 2.0)])
    pts = np.concatenate([near, far]).astype(np.float32)
    uv, cam, depth = points.project_pcd(pts, K, c2w)
    dm, idx = points.get_depth_map(uv, depth, h, w, scale=2)
    vis = points.mask_pcd_2d(uv, np.ones((h, w), np.float32), 0.5, dm, depth, depth_thresh=0.1)
    # (the near wall's rim blends with empty background cells in the bilinear lookup and drops out)
    assert vis[: near.shape[0]].mean() > 0.95 and vis[near.shape[0]:].mean() < 0.01
    winners = idx[idx < len(pts)]
    assert (winners < near.shape[0]).all() and len(winners) > 0.85 * idx.size
