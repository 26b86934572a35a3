"""Point-cloud z-buffer helpers on the GPU: the four functions of the
reference's `Articulation/utils/point_utils.py` with the same names, argument meaning and return
layout, over three HIP entry points of libmgs.so (mgs_points_project, mgs_points_depth_map,
mgs_points_sample_mask).  NumPy arrays in -> NumPy arrays out, like the reference; torch tensors
on the GPU in -> torch tensors out (no host round trip).  The camera here is the reference's
`c2w` used as is (camera looks down +z: point_utils.py:25), not the OpenGL convention of
`Camera.from_c2w_opengl`."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_handle


def _dev(x, device="cuda") -> torch.Tensor:
    t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    t = t.to(device=device, dtype=torch.float32).contiguous()
    if not t.is_cuda:
        raise _lib.MgsError("robosimgs_amd.points needs a GPU tensor / a visible GPU (no CPU fallback)")
    return t


def _like(x, t: torch.Tensor):
    return t if torch.is_tensor(x) else t.cpu().numpy()


def project_pcd(pnt_w, K, c2w):
    """point_utils.py:13-26.  pnt_w [N,3], K [3,3], c2w [4,4] -> (uv_cam [N,3], pnt_cam [N,3],
    depth [N,1])."""
    p, Kt, c = _dev(pnt_w).reshape(-1, 3), _dev(K).reshape(3, 3), _dev(c2w).reshape(4, 4)
    n = p.shape[0]
    uv = torch.empty(n, 3, dtype=torch.float32, device=p.device)
    cam = torch.empty(n, 3, dtype=torch.float32, device=p.device)
    check(_lib.lib().mgs_points_project(n, ptr(p), ptr(Kt), ptr(c), ptr(uv), ptr(cam), stream_handle()),
          "mgs_points_project")
    return _like(pnt_w, uv), _like(pnt_w, cam), _like(pnt_w, cam[:, 2:])


def unproject_pcd(pnt_cam, c2w):
    """point_utils.py:29-41: camera coordinates back to world, p = R pnt_cam + t (a 3x3 product;
    torch on the device)."""
    c, m = _dev(pnt_cam).reshape(-1, 3), _dev(c2w).reshape(4, 4)
    return _like(pnt_cam, c @ m[:3, :3].T + m[:3, 3])


def get_depth_map(uv, depth, h: int, w: int, bg_depth: float = 1e10, scale=2):
    """point_utils.py:44-73.  uv [N,2|3] pixel coordinates, depth [N] or [N,1] -> (depth_map
    [h,w] float32, index [int(w/scale) * int(h/scale)] int64: the point that won each low-resolution
    cell in the reference's u * _h + v order, N where none did)."""
    u = _dev(uv)
    u = u.reshape(-1, u.shape[-1])
    d = _dev(depth).reshape(-1)
    n = u.shape[0]
    if d.shape[0] != n:
        raise ValueError(f"{n} uv rows but {d.shape[0]} depths")
    _h, _w = int(h / scale), int(w / scale)
    if _h < 1 or _w < 1:
        raise ValueError(f"scale {scale} leaves no cells for a {w}x{h} image")
    depth_map = torch.empty(h, w, dtype=torch.float32, device=u.device)
    index = torch.empty(_w * _h, dtype=torch.int64, device=u.device)
    L = _lib.lib()
    nbytes = ctypes.c_size_t(0)
    args = [n, ptr(u), u.shape[1], ptr(d), int(h), int(w), _h, _w, float(scale), float(bg_depth),
            ptr(depth_map), ptr(index)]
    check(L.mgs_points_depth_map(*args, None, ctypes.byref(nbytes), stream_handle()),
          "mgs_points_depth_map(size query)")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=u.device)
    check(L.mgs_points_depth_map(*args, ptr(ws), ctypes.byref(nbytes), stream_handle()),
          "mgs_points_depth_map")
    return _like(uv, depth_map), _like(uv, index)


def mask_pcd_2d(uv, mask, thresh: float = 0.5, depth=None, pnt_depth=None,
                depth_thresh: float = 0.1):
    """point_utils.py:76-111.  uv [N,2|3], mask [H,W] (any numeric type), optional depth [H,W] +
    pnt_depth [N,1] -> bool [N]: the mask (bilinearly sampled) is above `thresh` at the point and,
    with a depth map, the point is within `depth_thresh` of the surface seen there."""
    u = _dev(uv)
    u = u.reshape(-1, u.shape[-1])
    m = _dev(mask)
    h, w = m.shape
    n = u.shape[0]
    dm = pd = None
    if depth is not None and pnt_depth is not None:
        dm, pd = _dev(depth), _dev(pnt_depth).reshape(n, -1)[:, 0].contiguous()
        if dm.shape != (h, w):
            raise ValueError("depth map and mask must share a shape")
    out = torch.empty(n, dtype=torch.uint8, device=u.device)
    check(_lib.lib().mgs_points_sample_mask(n, ptr(u), u.shape[1], ptr(m), h, w, float(thresh),
                                            ptr(dm), ptr(pd), float(depth_thresh), ptr(out),
                                            stream_handle()), "mgs_points_sample_mask")
    return _like(uv, out.bool())
