"""Similarity transforms of Gaussian groups on the GPU: world-frame alignment
of a whole scene, or -- every frame -- rigid motion of the Gaussians that ride on articulated
parts (a drawer, a lid, a robot link).  One HBM-bound HIP kernel (mgs_transform_gaussians);
`Gaussians.transformed()` in gaussians.py is the host-side NumPy counterpart for one transform."""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, require_device, stream_handle
from .gaussians import _rotmat_to_quat, sh_rotation_matrices
from .ops import _f32c


def pack_transforms(rotations: Sequence, translations: Sequence, scales: Optional[Sequence] = None,
                    sh_degree: int = 0):
    """Host side, per group: (xforms [G,20], sh_rot [G,84] | None) float32 arrays in the layout
    mgs_transform_gaussians reads.  rotations [G,3,3] proper rotations, translations [G,3],
    scales [G] uniform scale factors (default 1)."""
    R = np.asarray(rotations, dtype=np.float64).reshape(-1, 3, 3)
    t = np.asarray(translations, dtype=np.float64).reshape(-1, 3)
    G = R.shape[0]
    s = np.ones(G) if scales is None else np.asarray(scales, dtype=np.float64).reshape(G)
    if t.shape[0] != G:
        raise ValueError(f"{G} rotations but {t.shape[0]} translations")
    x = np.zeros((G, 20), dtype=np.float32)
    rot = np.zeros((G, 84), dtype=np.float32) if sh_degree >= 1 else None
    for k in range(G):
        if abs(np.linalg.det(R[k]) - 1.0) > 1e-4 or np.abs(R[k] @ R[k].T - np.eye(3)).max() > 1e-4:
            raise ValueError(f"group {k}: not a proper rotation (pass the scale separately)")
        x[k, :9] = (s[k] * R[k]).reshape(9)
        x[k, 9:12] = t[k]
        x[k, 12:16] = _rotmat_to_quat(R[k])
        x[k, 16] = s[k]
        if rot is not None:
            Ms = sh_rotation_matrices(R[k], sh_degree)
            off = 0
            for l in range(1, sh_degree + 1):
                m = 2 * l + 1
                rot[k, off:off + m * m] = Ms[l].reshape(-1)
                off += m * m
    return x, rot


def transform_gaussians(tensors: Dict, rotations=None, translations=None, scales=None,
                        group_ids: Optional[torch.Tensor] = None, rotate_sh: bool = True,
                        out: Optional[Dict] = None, packed=None) -> Dict:
    """Apply x -> s_g R_g x + t_g to the Gaussians of every group g.

    tensors: dict(means, quats, scales, opacities, colors [N,K,3], sh_degree) on the GPU
    (Gaussians.to_torch()).  group_ids: int32 [N] on the GPU, -1 = static; None = one group.
    Returns a new dict sharing `opacities`; pass out=<previous result> (or out=tensors for in
    place) to reuse buffers every frame.  rotate_sh=False leaves the view-dependent colour in
    the world frame (cheaper: the 192-byte SH rows are not touched).
    packed=(xforms [G,20], sh_rot [G,84] | None) device tensors from `pack_transforms` skip the
    host-side packing (about 45 us per group with SH matrices)."""
    require_device(tensors["means"], group_ids)
    deg = int(tensors.get("sh_degree") or 0)
    colors = tensors["colors"]
    do_sh = rotate_sh and colors.dim() == 3 and colors.shape[1] >= (deg + 1) ** 2 and deg >= 1
    dev = tensors["means"].device
    if packed is not None:
        xd, rd = packed
        if do_sh and rd is None:
            raise ValueError("packed transforms carry no SH matrices: pack with sh_degree or pass rotate_sh=False")
        n_groups = int(xd.shape[0])
    else:
        x, rot = pack_transforms(rotations, translations, scales, deg if do_sh else 0)
        xd = torch.from_numpy(x).to(dev)
        rd = torch.from_numpy(rot).to(dev) if rot is not None else None
        n_groups = x.shape[0]
    n = tensors["means"].shape[0]
    means, quats, scl = _f32c(tensors["means"]), _f32c(tensors["quats"]), _f32c(tensors["scales"])
    res = out if out is not None else {}
    o_means = res.get("means") if out is not None else None
    o_means = o_means if o_means is not None else torch.empty_like(means)
    o_quats = res.get("quats") if out is not None and res.get("quats") is not None else torch.empty_like(quats)
    o_scl = res.get("scales") if out is not None and res.get("scales") is not None else torch.empty_like(scl)
    sh_in = _f32c(colors) if do_sh else None
    o_sh = None
    if do_sh:
        o_sh = res.get("colors") if out is not None and res.get("colors") is not None else torch.empty_like(sh_in)
    gids = group_ids.to(torch.int32).contiguous() if group_ids is not None else None
    check(_lib.lib().mgs_transform_gaussians(
        n, ptr(means), ptr(quats), ptr(scl), deg, int(colors.shape[1]) if colors.dim() == 3 else 1, ptr(sh_in), ptr(gids), n_groups, ptr(xd),
        ptr(rd) if do_sh else None, ptr(o_means), ptr(o_quats), ptr(o_scl), ptr(o_sh), stream_handle()),
        "mgs_transform_gaussians")
    return {"means": o_means, "quats": o_quats, "scales": o_scl, "opacities": tensors["opacities"],
            "colors": o_sh if do_sh else colors, "sh_degree": tensors.get("sh_degree")}
