"""Compositing the rendered 3DGS background with a simulator / mesh render:
the "Holistic Scene Augmentation -> Simulated Data" step the reference names
(/root/reference/README.md:53-56) but has not released.  One HBM-bound HIP kernel
(mgs_composite_over); see include/mgs.h for the per-pixel rule."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check, ptr, require_device, stream_handle
from .ops import _f32c


def composite_over(bg_rgb: torch.Tensor, bg_alpha: torch.Tensor, bg_depth: torch.Tensor,
                   fg_rgb: torch.Tensor, fg_depth: torch.Tensor,
                   fg_mask: Optional[torch.Tensor] = None,
                   backdrop: Optional[Sequence[float]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """bg_rgb [...,3] (premultiplied accumulation as returned by `rasterization`), bg_alpha [...,1]
    or [...], bg_depth [...,1] or [...] (z-depth, the "ED" channel); fg_rgb [...,3], fg_depth [...]
    (z-depth of the opaque foreground; <= 0 or inf where absent unless fg_mask is given).
    Returns (rgb [...,3], depth [...])."""
    require_device(bg_rgb, bg_alpha, bg_depth, fg_rgb, fg_depth, fg_mask)
    lead = bg_rgb.shape[:-1]
    n_px = int(torch.Size(lead).numel())
    flat = lambda t_: _f32c(t_).reshape(n_px)
    b_rgb, f_rgb = _f32c(bg_rgb).reshape(n_px, 3), _f32c(fg_rgb).reshape(n_px, 3)
    b_a, b_d, f_d = flat(bg_alpha), flat(bg_depth), flat(fg_depth)
    mask = fg_mask.reshape(n_px).to(torch.uint8).contiguous() if fg_mask is not None else None
    bd = (torch.tensor(list(backdrop), dtype=torch.float32, device=bg_rgb.device)
          if backdrop is not None else None)
    out_rgb = torch.empty(n_px, 3, dtype=torch.float32, device=bg_rgb.device)
    out_depth = torch.empty(n_px, dtype=torch.float32, device=bg_rgb.device)
    check(_lib.lib().mgs_composite_over(n_px, ptr(b_rgb), ptr(b_a), ptr(b_d), ptr(f_rgb), ptr(f_d),
                                        ptr(mask), ptr(bd), ptr(out_rgb), ptr(out_depth),
                                        stream_handle()), "mgs_composite_over")
    return out_rgb.reshape(*lead, 3), out_depth.reshape(*lead)


def frame_to_u8(colors: torch.Tensor, alphas: torch.Tensor,
                background: Optional[Sequence[float]] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """colors [...,D>=3] (the first three channels are RGB, so an "RGB+ED" render works as is),
    alphas [...,1] or [...] -> uint8 [...,3] = round(255 * clamp(rgb + (1 - alpha) * bg, 0, 1)):
    the image a dataset writer stores, a quarter of the bytes to gather or download."""
    require_device(colors, alphas)
    lead = colors.shape[:-1]
    n_px = int(torch.Size(lead).numel())
    c = _f32c(colors).reshape(n_px, colors.shape[-1])
    a = _f32c(alphas).reshape(n_px)
    bg = (torch.tensor(list(background), dtype=torch.float32, device=colors.device)
          if background is not None else None)
    if out is None:
        out = torch.empty(n_px, 3, dtype=torch.uint8, device=colors.device)
    check(_lib.lib().mgs_frame_to_u8(n_px, ptr(c), c.shape[1], ptr(a), ptr(bg), ptr(out), stream_handle()),
          "mgs_frame_to_u8")
    return out.reshape(*lead, 3)
