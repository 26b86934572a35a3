"""oracle/points_np.py (restatement of the reference's utils/point_utils.py, parity unpinned)
against closed-form cases, and the C ABI's argument checks for the three point entry points."""
import numpy as np

from oracle import points_np as P


def _look(cam_pos, R=np.eye(3)):
    c2w = np.eye(4)
    c2w[:3, :3], c2w[:3, 3] = R, cam_pos
    return c2w


def test_project_is_a_plus_z_pinhole_and_unproject_inverts_it():
    K = np.array([[100.0, 0, 32], [0, 120.0, 24], [0, 0, 1]])
    c2w = _look([1.0, 2.0, 3.0])
    pts = np.array([[1.0, 2.0, 5.0], [2.0, 2.0, 5.0], [1.0, 3.0, 7.0]])
    uv, cam, depth = P.project_pcd(pts, K, c2w)
    np.testing.assert_allclose(cam, pts - [1, 2, 3])
    np.testing.assert_allclose(depth[:, 0], [2, 2, 4])
    np.testing.assert_allclose(uv, [[32, 24, 1], [32 + 100 / 2, 24, 1], [32, 24 + 120 / 4, 1]])
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    c2w = _look(rng.normal(size=3), q)
    pts = rng.normal(size=(50, 3))
    _, cam, _ = P.project_pcd(pts, K, c2w)
    np.testing.assert_allclose(P.unproject_pcd(cam, c2w), pts, atol=1e-12)


def test_depth_map_keeps_the_nearest_point_and_reports_the_first_winner():
    h, w = 8, 12
    uv = np.array([[4.0, 2.0], [4.4, 2.2], [5.0, 3.0], [100.0, -7.0], [4.0, 2.0]])   # cells at scale 2
    depth = np.array([3.0, 1.5, 9.0, 2.0, 1.5])
    dm, index = P.get_depth_map(uv, depth, h, w, bg_depth=50.0, scale=2)
    assert dm.shape == (h, w) and index.shape == (6 * 4,)
    # points 0, 1, 4 round to cell (2, 1): the minimum 1.5 is first reached by point 1
    assert index[2 * 4 + 1] == 1 and dm[2, 4] == dm[3, 5] == 1.5
    # point 2: round(2.5) = 2 (half to even), round(1.5) = 2 -> cell (2, 2)
    assert index[2 * 4 + 2] == 2 and dm[4, 4] == 9.0
    # the off-image point is clipped to the last column / first row, as in the reference
    assert index[5 * 4 + 0] == 3 and dm[0, 11] == 2.0
    assert (index == 5).sum() == 24 - 3 and dm[7, 0] == 50.0
    # a point at or beyond bg_depth never wins
    _, idx2 = P.get_depth_map(uv[:1], np.array([50.0]), h, w, bg_depth=50.0, scale=2)
    assert (idx2 == 1).all()


def test_mask_lookup_is_bilinear_with_border_padding():
    mask = np.zeros((4, 6), dtype=np.float32)
    mask[:, 3:] = 1.0
    # align_corners=True: u = w/2 maps to x = (w-1)/2 = 2.5 -> halfway between columns 2 and 3
    uv = np.array([[3.0, 2.0], [3.0 * 6 / 5 - 0.3, 2.0], [-40.0, 2.0], [40.0, 2.0]])
    assert np.isclose(P._grid_sample_bilinear(mask, uv)[0], 0.5)
    got = P.mask_pcd_2d(uv, mask, thresh=0.5)
    assert got.tolist() == [False, True, False, True]
    depth = np.full((4, 6), 2.0, dtype=np.float32)
    got = P.mask_pcd_2d(uv, mask, 0.5, depth, np.array([[2.05], [2.2], [2.0], [1.95]]), 0.1)
    assert got.tolist() == [False, False, False, True]


def test_point_entry_points_reject_bad_arguments_without_a_gpu():
    from robosimgs_amd import _lib
    import ctypes
    L = _lib.lib()
    assert L.mgs_points_project(-1, None, None, None, None, None, None) == -1
    nb = ctypes.c_size_t(0)
    assert L.mgs_points_depth_map(0, None, 2, None, 8, 12, 4, 6, 2.0, 1e10, None, None, None,
                                  ctypes.byref(nb), None) == 0 and nb.value == 4 * 6 * 8
    assert L.mgs_points_depth_map(4, None, 1, None, 8, 12, 4, 6, 2.0, 1e10, None, None, None,
                                  ctypes.byref(nb), None) == -1
    assert b"uv_stride" in L.mgs_last_error_string()
    assert L.mgs_points_sample_mask(0, None, 2, None, 4, 4, 0.5, None, None, 0.1, None, None) == 0
