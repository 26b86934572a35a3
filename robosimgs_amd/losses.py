"""Loss terms of the training step as fused HIP kernels.  `l1_loss(render, target)` is
`(render - target).abs().mean()` with its gradient.  Where the render wants a gradient the forward is ONE streaming
launch that also leaves sign(render - target) / n (mgs_l1_loss_fwd_grad) and the backward a launch that does nothing
for the usual grad_output of 1 (mgs_l1_loss_bwd_scale); without a gradient, mgs_l1_loss_fwd alone.  Instead of six
elementwise / reduction kernels of eager PyTorch; bit-reproducible."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import check, ptr, require_device, stream_handle
from .ops import _f32c


_ONES: dict = {}


def unit_gradient(loss: torch.Tensor) -> torch.Tensor:
    """A cached scalar 1.0 on `loss`'s device for `loss.backward(gradient=unit_gradient(loss))`: autograd then needs no
    `ones_like` fill launch, and `l1_loss`'s backward, handed this very tensor, knows the cotangent is 1 without reading it
    and skips its scale launch -- two launches less per training step (`Trainer.step` does this).  Never written to."""
    key = (loss.device.type, loss.device.index)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), dtype=torch.float32, device=loss.device)
    return one


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        n = a.numel()
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        L = _lib.lib()
        nbytes = ctypes.c_size_t(0)
        if not ctx.needs_input_grad[0]:
            check(L.mgs_l1_loss_fwd(n, None, None, None, None, ctypes.byref(nbytes), None), "mgs_l1_loss_fwd(size query)")
            ws = torch.empty(nbytes.value, dtype=torch.uint8, device=a.device)
            check(L.mgs_l1_loss_fwd(n, ptr(a), ptr(b), ptr(loss), ptr(ws), ctypes.byref(nbytes),
                                    stream_handle()), "mgs_l1_loss_fwd")
            return loss
        check(L.mgs_l1_loss_fwd_grad(n, None, None, None, None, None, ctypes.byref(nbytes), None), "mgs_l1_loss_fwd_grad(size query)")
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=a.device)
        v_a = torch.empty_like(a)
        check(L.mgs_l1_loss_fwd_grad(n, ptr(a), ptr(b), ptr(loss), ptr(v_a), ptr(ws), ctypes.byref(nbytes),
                                     stream_handle()), "mgs_l1_loss_fwd_grad")
        ctx.save_for_backward(a, b)
        ctx.v_a = v_a
        return loss

    @staticmethod
    def backward(ctx, v_loss):
        g = _f32c(v_loss)
        v_a, ctx.v_a = ctx.v_a, None
        if v_a is not None:
            one = _ONES.get((g.device.type, g.device.index))
            if one is None or g.data_ptr() != one.data_ptr():      # (unit_gradient's tensor: the cotangent is 1, nothing to scale)
                check(_lib.lib().mgs_l1_loss_bwd_scale(v_a.numel(), ptr(g), ptr(v_a), stream_handle()), "mgs_l1_loss_bwd_scale")
            return v_a, None
        # a second backward through a retained graph: autograd owns the first buffer by now
        a, b = ctx.saved_tensors
        v_a = torch.empty_like(a)
        check(_lib.lib().mgs_l1_loss_bwd(a.numel(), ptr(a), ptr(b), ptr(g), ptr(v_a), stream_handle()), "mgs_l1_loss_bwd")
        return v_a, None


def l1_loss(render: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """mean |render - target| (scalar tensor); gradient flows to `render` only."""
    if render.shape != target.shape:
        raise ValueError(f"shape mismatch {tuple(render.shape)} vs {tuple(target.shape)}")
    require_device(render, target)
    a, b = _f32c(render), _f32c(target.detach())
    if a.data_ptr() % 16:            # a contiguous view at an odd offset: the kernels load float4
        a = a.clone()
    if b.data_ptr() % 16:
        b = b.clone()
    return _L1.apply(a, b)
