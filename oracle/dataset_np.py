"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of mgs_frame_to_dataset (robosimgs_amd/csrc/composite.hip).

The layout it produces is defined by the reference's READERS, which tests/golden/make_dataset_golden.py runs
on files made from this restatement:
  load_images   /root/reference/Articulation/utils/nerf2physic_utils.py:84-101   RGBA, mask = alpha > 0
  load_depths   :104-118  [H,W,1] ray distance in .npy.gz  ->  distance_to_depth (:135-146)
  depth_to_distance :120-132  distance = z * ||K^-1 (x, y, 1)||, x / y integer pixel coordinates.
Parity PINNED by tests/golden/dataset_reference.npz (what those readers returned).  Only tests/ and the
golden generator import this module; the product never does.
"""
import numpy as np


def quant8(v):
    """round(255 * clamp(v, 0, 1)) with fp32 arithmetic and round-half-to-even (v_cvt / __float2int_rn)."""
    v = np.clip(np.asarray(v, dtype=np.float32), np.float32(0), np.float32(1))
    return np.rint(np.float32(255.0) * v).astype(np.uint32)


def frame_to_dataset(colors, alphas, K=None, background=None, distance_f64=False):
    """colors [H,W,D] fp32, alphas [H,W] fp32 -> (rgba uint8 [H,W,4], distance [H,W,1] or None)."""
    c = np.asarray(colors, dtype=np.float32)
    a = np.asarray(alphas, dtype=np.float32).reshape(c.shape[:2])
    h, w, _ = c.shape
    bg = np.zeros(3, np.float32) if background is None else np.asarray(background, dtype=np.float32)[:3]
    wgt = (np.float32(1.0) - a)[..., None]
    rgb = quant8(c[..., :3] + wgt * bg)                         # fp32: multiply, then add (no fused multiply-add)
    A = np.where(a > 0, np.maximum(1, quant8(a)), 0).astype(np.uint32)
    rgba = np.concatenate([rgb, A[..., None]], axis=-1).astype(np.uint8)
    dist = None
    if K is not None:
        ki = np.linalg.inv(np.asarray(K, dtype=np.float64).reshape(3, 3))
        x, y = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        rx = (x * ki[0, 0] + y * ki[0, 1]) + ki[0, 2]
        ry = (x * ki[1, 0] + y * ki[1, 1]) + ki[1, 2]
        rz = (x * ki[2, 0] + y * ki[2, 1]) + ki[2, 2]
        norm = np.sqrt((rx * rx + ry * ry) + rz * rz)
        d64 = c[..., -1].astype(np.float64) * norm
        dist = (d64 if distance_f64 else d64.astype(np.float32))[..., None]
    return rgba, dist
