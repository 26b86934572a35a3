"""Pinhole cameras in the conventions RoboSimGS uses.

The reference stores cameras as OpenGL camera-to-world matrices (+X right, +Y up, camera
looks down -Z) plus a 3x3 intrinsic matrix:
  * nerfstudio `transforms.json` frames (`transform_matrix`, `fl_x`, `fl_y`, `cx`, `cy`)
    -- /root/reference/Articulation/utils/nerf2physic_utils.py:26-52
  * `camera_params.json` written by the segmenter (`intrinsics`, `c2w`, `resolution`)
    -- /root/reference/Articulation/segmentation/interactive_segmenter.py:279-320
The renderer consumes OpenCV world-to-camera view matrices (+Z forward, +Y down).  The
conversion is the one `project_3d_to_2d` applies (nerf2physic_utils.py:14-16): invert
c2w, negate camera Y and Z.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np

_FLIP_YZ = np.diag([1.0, -1.0, -1.0, 1.0])


@dataclass
class Camera:
    """One pinhole camera.  `c2w` is OpenGL camera-to-world (4x4, float64)."""

    c2w: np.ndarray
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int
    near: float = 0.01
    far: float = 1e10

    def __post_init__(self):
        self.c2w = np.asarray(self.c2w, dtype=np.float64).reshape(4, 4)
        self.width = int(self.width)
        self.height = int(self.height)

    # -- constructors --------------------------------------------------------------
    @classmethod
    def from_c2w_opengl(cls, c2w, K, width, height, **kw) -> "Camera":
        K = np.asarray(K, dtype=np.float64)
        return cls(c2w, K[0, 0], K[1, 1], K[0, 2], K[1, 2], width, height, **kw)

    @classmethod
    def from_w2c_opencv(cls, viewmat, K, width, height, **kw) -> "Camera":
        """Inverse of :meth:`viewmat`."""
        c2w = np.linalg.inv(_FLIP_YZ @ np.asarray(viewmat, dtype=np.float64))
        return cls.from_c2w_opengl(c2w, K, width, height, **kw)

    @classmethod
    def from_camera_params(cls, entry: Dict, **kw) -> "Camera":
        """One entry of the segmenter's `camera_params.json`
        (interactive_segmenter.py:313-320: keys intrinsics / c2w / resolution)."""
        w, h = entry["resolution"]
        return cls.from_c2w_opengl(entry["c2w"], entry["intrinsics"], w, h, **kw)

    @classmethod
    def look_at(cls, position, target, up, width, height, fov_x_deg, **kw) -> "Camera":
        """Look-at construction with the reference's column layout
        (interactive_segmenter.py:291-311: columns right / up / -forward / position)
        and its fov -> focal rule (:280-281, applied to the image width)."""
        position = np.asarray(position, dtype=np.float64)
        forward = np.asarray(target, dtype=np.float64) - position
        forward /= np.linalg.norm(forward)
        right = np.cross(forward, np.asarray(up, dtype=np.float64))
        right /= np.linalg.norm(right)
        true_up = np.cross(right, forward)
        true_up /= np.linalg.norm(true_up)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, true_up, -forward, position
        f = (width / 2.0) / math.tan(math.radians(fov_x_deg / 2.0))
        return cls(c2w, f, f, width / 2.0, height / 2.0, width, height, **kw)

    # -- derived quantities --------------------------------------------------------
    @property
    def K(self) -> np.ndarray:
        return np.array([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy],
                         [0.0, 0.0, 1.0]])

    def viewmat(self) -> np.ndarray:
        """OpenCV world-to-camera 4x4: diag(1,-1,-1,1) @ inv(c2w)."""
        return _FLIP_YZ @ np.linalg.inv(self.c2w)

    @property
    def position(self) -> np.ndarray:
        return self.c2w[:3, 3].copy()

    def project(self, pts: np.ndarray, return_dists: bool = False):
        """Pixel coordinates of world points; same contract as the reference's
        `project_3d_to_2d(pts, w2c, K, return_dists)` (nerf2physic_utils.py:10-23)."""
        pts = np.asarray(pts, dtype=np.float64)
        pc = pts @ self.viewmat()[:3, :3].T + self.viewmat()[:3, 3]
        uvw = pc @ self.K.T
        uv = uvw[:, :2] / uvw[:, 2:]
        if return_dists:
            return uv, np.linalg.norm(pc, axis=-1)
        return uv

    def scaled(self, factor: float) -> "Camera":
        """Same view at a different resolution."""
        return Camera(self.c2w.copy(), self.fx * factor, self.fy * factor,
                      self.cx * factor, self.cy * factor,
                      int(round(self.width * factor)), int(round(self.height * factor)),
                      self.near, self.far)


def cameras_from_transforms_json(path: str, width: int | None = None,
                                 height: int | None = None) -> List[Camera]:
    """nerfstudio `transforms.json` -> cameras.  Accepts global or per-frame intrinsics,
    the two layouts `parse_transforms_json` reads (nerf2physic_utils.py:30-45)."""
    with open(path, "rb") as f:
        t = json.load(f)
    cams = []
    for fr in t["frames"]:
        src = fr if "fl_x" in fr else t
        w = int(src.get("w", t.get("w", width or round(2 * src["cx"]))))
        h = int(src.get("h", t.get("h", height or round(2 * src["cy"]))))
        cams.append(Camera(fr["transform_matrix"], src["fl_x"], src["fl_y"], src["cx"],
                           src["cy"], w, h))
    return cams


def cameras_from_camera_params_json(path: str) -> Dict[str, Camera]:
    with open(path, "r") as f:
        d = json.load(f)
    return {k: Camera.from_camera_params(v) for k, v in d.items()}


def camera_ring(n: int, width: int, height: int, radius: float = 7.0,
                height_z: float = 1.5, fov_x_deg: float = 60.0,
                thetas: Sequence[float] | None = None) -> List[Camera]:
    """Look-at ring used by every benchmark config: position
    (r cos t, r sin t, h), target origin, world up +Z."""
    if thetas is None:
        thetas = [2.0 * math.pi * k / n for k in range(n)]
    return [Camera.look_at((radius * math.cos(t), radius * math.sin(t), height_z),
                           (0.0, 0.0, 0.0), (0.0, 0.0, 1.0), width, height, fov_x_deg)
            for t in thetas]


def depth_to_distance(depth: np.ndarray, K: np.ndarray) -> np.ndarray:
    """z-depth map -> ray distance, |K^-1 [u,v,1]| * depth at integer pixel (u,v)
    (same contract as nerf2physic_utils.py:120-132)."""
    h, w = depth.shape
    return depth * _ray_norm(h, w, K)


def distance_to_depth(dists: np.ndarray, K: np.ndarray) -> np.ndarray:
    """Inverse of :func:`depth_to_distance` (nerf2physic_utils.py:135-146)."""
    h, w = dists.shape
    return dists / _ray_norm(h, w, K)


def _ray_norm(h: int, w: int, K: np.ndarray) -> np.ndarray:
    Kinv = np.linalg.inv(np.asarray(K, dtype=np.float64))
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    rays = np.stack([u, v, np.ones_like(u)], axis=-1) @ Kinv.T
    return np.linalg.norm(rays, axis=-1)


def unproject_point(pt_2d, depth: np.ndarray, c2w: np.ndarray, K: np.ndarray):
    """Pixel + z-depth map -> world point in the OpenGL camera convention
    (nerf2physic_utils.py:172-185: camera-space ray [x, -y, -1] * depth)."""
    K = np.asarray(K, dtype=np.float64)
    x = (pt_2d[0] - K[0, 2]) / K[0, 0]
    y = (pt_2d[1] - K[1, 2]) / K[1, 1]
    p = np.array([x, -y, -1.0]) * depth[pt_2d[1], pt_2d[0]]
    return (np.asarray(c2w, dtype=np.float64) @ np.append(p, 1.0))[:3]
