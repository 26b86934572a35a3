"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's point-cloud z-buffer helpers
(SURVEY.md 8(f4)).  Only tests/ may import this module; the product path is
robosimgs_amd/points.py over libmgs.so.

PARITY UNPINNED: /root/reference/Articulation/utils/point_utils.py cannot be imported here (its
module header needs torch_scatter and cv2, neither is installed) and the reference holds no test
or fixture for it, so this file restates the four functions from their source text and is checked
against closed-form cases only (tests/test_points_oracle.py).

  project_pcd      point_utils.py:13-26   pnt_cam = R^T (p - t) with c2w = [R | t];  uv = (pnt_cam / z) K^T
  unproject_pcd    point_utils.py:29-41   p = R pnt_cam + t
  get_depth_map    point_utils.py:44-73   scatter-min z-buffer at 1/scale resolution, nearest upsample
  mask_pcd_2d      point_utils.py:76-111  bilinear mask / depth lookup (grid_sample, align_corners=True,
                                          border padding) -> per-point visibility

Two library behaviours are restated from their published semantics:
  * torch_scatter.scatter_min(src, index, out=out): out[i] = min(out[i], min of src at i); the arg
    output is the position of the FIRST minimal src element, or len(src) where no element is
    strictly below the initial out value.
  * cv2.resize(..., interpolation=INTER_NEAREST): dst[y, x] = src[min(floor(y * sh / dh), sh - 1),
    min(floor(x * sw / dw), sw - 1)].
"""
import numpy as np


def project_pcd(pnt_w, K, c2w):
    """point_utils.py:13-26.  Returns (uv_cam [N,3] = (u, v, 1), pnt_cam [N,3], depth [N,1])."""
    pnt_w, K, c2w = (np.asarray(a, dtype=np.float64) for a in (pnt_w, K, c2w))
    pnt_cam = (pnt_w - c2w[:3, 3]) @ c2w[:3, :3]           # rows: R^T (p - t)
    uv_cam = (pnt_cam / pnt_cam[..., 2:]) @ K.T
    return uv_cam, pnt_cam, pnt_cam[..., 2:]


def unproject_pcd(pnt_cam, c2w):
    """point_utils.py:29-41."""
    pnt_cam, c2w = np.asarray(pnt_cam, dtype=np.float64), np.asarray(c2w, dtype=np.float64)
    return (c2w[:3, :3] @ pnt_cam.T + c2w[:3, 3, None]).T


def get_depth_map(uv, depth, h, w, bg_depth=1e10, scale=2):
    """point_utils.py:44-73.  Returns (depth_map [h,w] float32, index [(w/scale)*(h/scale)] int64
    in the reference's column-major cell order u * _h + v; index == N where no point won)."""
    _h, _w = int(h / scale), int(w / scale)
    uv = np.round(np.asarray(uv, dtype=np.float64) / scale).astype(np.int32)[..., :2]   # half to even
    uv = uv.clip(0, np.array([_w, _h]) - 1)
    d = np.asarray(depth, dtype=np.float32).reshape(-1)
    cell = uv[:, 0].astype(np.int64) * _h + uv[:, 1]
    n = d.shape[0]
    cells = np.full(_w * _h, np.float32(bg_depth), dtype=np.float32)
    index = np.full(_w * _h, n, dtype=np.int64)
    for i in range(n):                                     # sequential scatter-min, first minimum wins
        c = cell[i]
        if d[i] < cells[c]:
            cells[c] = d[i]
            index[c] = i
    small = cells.reshape(_w, _h).T                        # [_h, _w]
    ys = np.minimum(np.arange(h) * _h // h, _h - 1)        # floor(y * _h / h) in exact arithmetic
    xs = np.minimum(np.arange(w) * _w // w, _w - 1)
    return small[ys][:, xs], index


def _grid_sample_bilinear(img, uv):
    """F.grid_sample(img[None,None], grid, padding_mode='border', align_corners=True) at
    normalised coordinates (uv - [w/2, h/2]) / [w/2, h/2]  (point_utils.py:92-101)."""
    h, w = img.shape
    img = np.asarray(img, dtype=np.float32)
    uv = np.asarray(uv, dtype=np.float64)[:, :2]
    gx = ((uv[:, 0] - w / 2) / (w / 2)).astype(np.float32)
    gy = ((uv[:, 1] - h / 2) / (h / 2)).astype(np.float32)
    x = np.clip((gx + 1) / 2 * np.float32(w - 1), 0, w - 1).astype(np.float32)
    y = np.clip((gy + 1) / 2 * np.float32(h - 1), 0, h - 1).astype(np.float32)
    x0, y0 = np.floor(x), np.floor(y)
    fx, fy = x - x0, y - y0
    x0i, y0i = x0.astype(np.int64), y0.astype(np.int64)
    x1i, y1i = np.minimum(x0i + 1, w - 1), np.minimum(y0i + 1, h - 1)
    return (img[y0i, x0i] * (1 - fx) * (1 - fy) + img[y0i, x1i] * fx * (1 - fy)
            + img[y1i, x0i] * (1 - fx) * fy + img[y1i, x1i] * fx * fy).astype(np.float32)


def mask_pcd_2d(uv, mask, thresh=0.5, depth=None, pnt_depth=None, depth_thresh=0.1):
    """point_utils.py:76-111.  Returns bool [N]."""
    out = _grid_sample_bilinear(np.asarray(mask, dtype=np.float32), uv) > thresh
    if depth is not None and pnt_depth is not None:
        sd = _grid_sample_bilinear(np.asarray(depth, dtype=np.float32), uv)
        out = out & (np.abs(sd - np.asarray(pnt_depth, dtype=np.float32).reshape(len(sd), -1)[:, 0])
                     < depth_thresh)
    return out
