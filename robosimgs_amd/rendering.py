"""`rasterization(...)`: the whole background-render path behind the signature nerfstudio's
splatfacto calls (gsplat 1.x `rasterization`), plus `render(...)`,
the Camera/Gaussians convenience the data-generation loop would use.

Per camera the frame is four C-ABI calls on torch's current stream:
    mgs_project_color_fwd -> mgs_isect_tiles -> mgs_rasterize_fwd        (forward)
    mgs_rasterize_bwd -> mgs_project_color_bwd                            (backward)
With `isect_capacity` given nothing is read back from the device, so a frame (or a training
step) can be captured in a HIP graph.  Without it the intersection bound is read back once
per camera to size the lists, as the reference operator does.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib, ops
from ._lib import check, ptr, require_device, stream_handle
from .ops import TILE_SIZE, _f32c

_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")


class _Meta(dict):
    """The `meta` dict of `rasterization`.  The render path keeps the per-camera tile lists on
    the device at their capacity; gsplat's flat `flatten_ids` / `isect_ids` tensors are
    materialised only if somebody asks for them (one read-back of the intersection counts)."""

    def __missing__(self, key):
        if key not in ("flatten_ids", "isect_ids") or "tile_lists" not in self:
            raise KeyError(key)
        lists = self["tile_lists"]
        if lists and lists[0].tile_ids is None:
            raise KeyError(f"{key}: this frame was rendered with lean_meta=True and keeps no tile ids")
        n_tiles = self["tile_width"] * self["tile_height"]
        tile_bits = int(n_tiles).bit_length()             # floor(log2(n_tiles)) + 1
        N = self["depths"].shape[1]
        counts = [int(c) for c in self["n_isects"].tolist()]
        flat, keys = [], []
        for c, (tl, n) in enumerate(zip(lists, counts)):
            ids = tl.flatten_ids[:n]
            flat.append(ids + c * N)
            depth_bits = self["depths"][c][ids.long()].view(torch.int32).long() & 0xFFFFFFFF
            keys.append((((c << tile_bits) | tl.tile_ids[:n].long()) << 32) | depth_bits)
        self["flatten_ids"] = torch.cat(flat)
        self["isect_ids"] = torch.cat(keys)
        return self[key]


class _RenderSH(torch.autograd.Function):
    """SH-coloured frames for C cameras: fused projection+colour, binning, raster."""

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, sh_coeffs, viewmats, Ks, backgrounds,
                width, height, sh_degree, eps2d, near_plane, far_plane, radius_clip,
                antialiased, with_depth, isect_capacity, absgrad, meta_out, tight, expected_depth,
                latency, lean, segment, dataset=None):
        C = viewmats.shape[0]
        dev = means.device
        # `tight` carries the binning policy: bit 0 tightened tile rectangles, bit 1 the per-axis (gsplat >= 1.5) radius rule
        per_axis, tight = bool(int(tight) & 2), bool(int(tight) & 1)
        tile_w, tile_h = -(-width // TILE_SIZE), -(-height // TILE_SIZE)
        ch = 4 if with_depth else 3
        render = torch.empty(C, height, width, ch, dtype=torch.float32, device=dev)
        alphas = torch.empty(C, height, width, dtype=torch.float32, device=dev)
        # last_ids (and the backward's slot map) only when some input wants a gradient
        training = any(ctx.needs_input_grad[:6]) or ctx.needs_input_grad[7]
        last_ids = (torch.empty(C, height, width, dtype=torch.int32, device=dev) if training
                    else None)
        # lean: an inference frame with a fixed list capacity keeps only what its own kernels read -- the packed
        # records, the binning seed and the depths; radii / means2d / conics / feats / tiles_per_gauss are neither
        # written nor returned (36 + 4 MB of stores per 1 M Gaussians)
        lean = bool(lean) and not training and isect_capacity is not None
        if dataset is not None and not lean:
            raise ValueError("dataset_out needs inference frames through the one-call path: lean_meta=True, isect_capacity "
                             "given, no gradients")
        if lean:
            # the whole batch of cameras behind ONE C call (mgs_render_frames): per-camera scratch is reused, nothing
            # per Gaussian is returned
            _, _, n_isects, status = ops.render_frames_raw(
                means, quats, scales, opacities, sh_degree, sh_coeffs, viewmats, Ks, width, height, eps2d,
                near_plane, far_plane, radius_clip, antialiased, with_depth, isect_capacity, backgrounds=backgrounds,
                expected_last=expected_depth, latency=latency, out=(render, alphas), tight=tight, per_axis=per_axis,
                dataset=dataset[:3] if dataset is not None else None,
                float_frame=dataset is None or bool(dataset[3]))
            meta_out["lean"] = dict(n_isects=n_isects, isect_status=status)
            ctx.set_materialize_grads(False)
            return render, alphas.unsqueeze(-1)
        if training and isect_capacity is not None:
            # the whole batch of training cameras behind ONE C call (mgs_render_frames_train): what the backward and the
            # meta dict need stays per camera in one state buffer; no read-back, so the step captures in a HIP graph
            _, _, st = ops.render_frames_train_raw(
                means, quats, scales, opacities, sh_degree, sh_coeffs, viewmats, Ks, width, height, eps2d, near_plane,
                far_plane, radius_clip, antialiased, with_depth, isect_capacity, segment, backgrounds=backgrounds,
                expected_last=expected_depth, latency=latency, tight=tight, out=(render, alphas), per_axis=per_axis)
            per_cam = []
            for c in range(C):
                v = st.views(c)
                per_cam.append((torch.stack([v["radii"], v["radii_y"]]) if per_axis else v["radii"], v["means2d"], v["depths"], v["conics"], v["opac_aa"] if antialiased else None,
                                v["feats"], st.tile_lists(c, v), v["splats"], None))
            ctx.train_state = st
            ctx.expected_depth = bool(expected_depth)
            ctx.channels = ch
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(means, quats, scales, opacities, sh_coeffs, viewmats, Ks, backgrounds, alphas, None, render)
            ctx.cfg = (width, height, tile_w, tile_h, sh_degree, eps2d, antialiased, with_depth, absgrad)
            meta_out["per_cam"] = per_cam
            ctx.meta_out = meta_out
            return render, alphas.unsqueeze(-1)
        ctx.train_state = None
        per_cam = []
        for c in range(C):
            # the projection kernel also seeds the binning (tile rectangle + count per Gaussian)
            radii, means2d, depths, conics, opac_aa, feats, splats, seed = ops.project_color_fwd_raw(
                means, quats, scales, opacities, sh_degree, sh_coeffs, viewmats[c], Ks[c], width,
                height, eps2d, near_plane, far_plane, radius_clip, antialiased, with_depth,
                want_splats=True, bin_seed="tight" if tight else "classic", lean=lean, per_axis=per_axis)
            opac = opac_aa if antialiased else opacities
            cap = isect_capacity
            if cap is None:
                cap = max(1, ops._upper_bound_isects(radii, tile_w, tile_h))
            tl = ops.isect_tiles_raw(means2d, radii, depths, tile_w, tile_h, cap, c, C,
                                     want_isect_ids=False, want_tiles_per_gauss=not lean,
                                     want_pair_info=training, want_tile_ids=not lean,
                                     conics=conics if tight else None,
                                     opacities=opac if tight else None, seed=seed, splats=splats if training else None)
            # training: the forward leaves per-pixel checkpoints every `segment` list entries, so that the backward
            # can walk a tile's list as independent segments (include/mgs.h: mgs_rasterize_fwd)
            # (without a fixed capacity `cap` is the loose bound read back above -- several times the lists: the checkpoints
            #  are then sized by the count the binning has just written, one more read-back on a path that reads back anyway)
            ckpt_cap = cap if isect_capacity is not None else min(cap, int(tl.n_isect.item()) + 1)
            ckpt = ops.checkpoint_buffer(ckpt_cap, tile_w, tile_h, ch, segment, dev) if (training and segment) else None
            ops.rasterize_fwd_raw(means2d, conics, feats, opac,
                                  backgrounds[c] if backgrounds is not None else None, width,
                                  height, tile_w, tile_h, tl.tile_offsets, tl.flatten_ids,
                                  out=(render[c], alphas[c], last_ids[c] if training else None),
                                  splats=splats, expected_last=expected_depth, latency=latency,
                                  group_order=tl.group_order, channels=ch, checkpoints=ckpt,
                                  checkpoint_interval=segment if ckpt is not None else 0)
            per_cam.append((radii, means2d, depths, conics, opac_aa, feats, tl, splats, ckpt))
        ctx.per_cam = per_cam
        # "RGB+ED": the kernel's epilogue divided the depth channel by max(alpha, 1e-10); the
        # backward undoes that with the saved frame (an OUTPUT: it must go through
        # save_for_backward -- parked on ctx it forms a reference cycle that crashes HIP graph capture)
        ctx.expected_depth = bool(expected_depth)
        ctx.channels = ch
        ctx.segment = int(segment) if training else 0
        ctx.set_materialize_grads(False)       # an unused output's cotangent arrives as None, not as a zero frame
        ctx.save_for_backward(means, quats, scales, opacities, sh_coeffs, viewmats, Ks,
                              backgrounds, alphas, last_ids,
                              render if ((expected_depth or segment) and training) else None)
        ctx.cfg = (width, height, tile_w, tile_h, sh_degree, eps2d, antialiased, with_depth,
                   absgrad)
        meta_out["per_cam"] = per_cam
        ctx.meta_out = meta_out
        return render, alphas.unsqueeze(-1)

    @staticmethod
    def backward(ctx, v_render, v_alphas):
        (means, quats, scales, opacities, sh_coeffs, viewmats, Ks, backgrounds, alphas,
         last_ids, render_out) = ctx.saved_tensors
        (width, height, tile_w, tile_h, sh_degree, eps2d, antialiased, with_depth,
         absgrad) = ctx.cfg
        C, n = viewmats.shape[0], means.shape[0]
        # set_materialize_grads(False): an output the loss does not use arrives as None instead of a zero frame
        if v_render is None:
            v_render = torch.zeros(C, height, width, ctx.channels, dtype=torch.float32, device=means.device)
        v_render = _f32c(v_render)
        v_alphas = _f32c(v_alphas).reshape(C, height, width) if v_alphas is not None else None
        if ctx.train_state is not None:
            # the batch's backward behind one C call (mgs_render_frames_backward): same kernels, same order
            v_means, v_quats, v_scales, v_sh, v_opacities, v_viewmats, v_m2d, v_abs = ops.render_frames_backward_raw(
                means, quats, scales, opacities, sh_degree, sh_coeffs, viewmats, Ks, eps2d, backgrounds, ctx.train_state,
                render_out, alphas, v_render, v_alphas, absgrad=absgrad, want_viewmats=ctx.needs_input_grad[5])
            ctx.meta_out["means2d_grad"] = [v_m2d[c] for c in range(C)]
            if absgrad:
                ctx.meta_out["means2d_absgrad"] = [v_abs[c] for c in range(C)]
            m2d = ctx.meta_out.get("means2d")
            if m2d is not None and m2d.requires_grad:
                m2d.grad = v_m2d
                if absgrad:
                    m2d.absgrad = v_abs
            v_bg = None
            if backgrounds is not None and ctx.needs_input_grad[7]:
                vr = v_render
                if ctx.expected_depth:
                    vr = torch.cat([vr[..., :-1], (vr[..., -1] / alphas.clamp(min=1e-10)).unsqueeze(-1)], dim=-1)
                v_bg = (vr * (1.0 - alphas).unsqueeze(-1)).sum(dim=(1, 2))
            return (v_means, v_quats, v_scales, v_opacities, v_sh, v_viewmats, None, v_bg) + (None,) * 18
        # "RGB+ED": the raster backward's prologue undoes the divide by max(alpha, 1e-10) itself
        # (expected_render=...); only a background gradient needs the converted cotangent here
        # the first camera overwrites the outputs, later ones accumulate: no zero-fill pass
        v_means = torch.empty_like(means)
        v_quats = torch.empty_like(quats)
        v_scales = torch.empty_like(scales)
        v_sh = torch.empty_like(sh_coeffs)
        v_opacities = torch.empty_like(opacities) if antialiased else None
        # camera-pose gradients only when asked for (float atomics into a zeroed [C,4,4])
        v_viewmats = torch.zeros_like(viewmats) if ctx.needs_input_grad[5] else None
        L = _lib.lib()
        for c in range(C):
            radii, means2d, depths, conics, opac_aa, feats, tl, splats, ckpt = ctx.per_cam[c]
            opac = opac_aa if antialiased else opacities
            bg = backgrounds[c] if backgrounds is not None else None
            v_means2d, v_conics, v_feats, v_opac, v_abs = ops.rasterize_bwd_det_raw(
                means2d, conics, feats, opac, bg, width, height, tile_w, tile_h, tl, alphas[c],
                last_ids[c], v_render[c], v_alphas[c] if v_alphas is not None else None, absgrad, splats=splats,
                expected_render=render_out[c] if ctx.expected_depth else None,
                render_out=render_out[c] if ckpt is not None else None, checkpoints=ckpt,
                checkpoint_interval=ctx.segment if ckpt is not None else 0)
            # screen-space gradients for densification strategies (gsplat exposes them through
            # means2d.grad / means2d.absgrad; here they are published in the meta dict)
            ctx.meta_out.setdefault("means2d_grad", [None] * C)[c] = v_means2d
            # (all four blend-stage gradients, for tests that gate the raster backward on its own)
            ctx.meta_out.setdefault("blend_grads", [None] * C)[c] = (v_means2d, v_conics, v_feats, v_opac)
            if absgrad:
                ctx.meta_out.setdefault("means2d_absgrad", [None] * C)[c] = v_abs
            check(L.mgs_project_color_bwd(
                n, ptr(means), ptr(quats), ptr(scales), ptr(opacities), sh_degree,
                sh_coeffs.shape[1], ptr(sh_coeffs), ptr(viewmats[c]), ptr(Ks[c]), width, height,
                eps2d, ptr(ops.radii_x(radii)), ptr(conics), int(antialiased), feats.shape[1], ptr(feats),
                ptr(v_feats), ptr(v_means2d), ptr(v_conics), None,
                ptr(v_opac) if antialiased else None, ptr(v_means), ptr(v_quats), ptr(v_scales),
                ptr(v_sh), ptr(v_opacities),
                ptr(v_viewmats[c]) if v_viewmats is not None else None, int(c > 0),
                stream_handle()),
                "mgs_project_color_bwd")
            if not antialiased:
                v_opacities = v_opac if v_opacities is None else v_opacities + v_opac
        # gsplat users call meta["means2d"].retain_grad() and read .grad / .absgrad after backward
        # (densification).  meta["means2d"] is handed out as a leaf that receives them here.
        m2d = ctx.meta_out.get("means2d")
        if m2d is not None and m2d.requires_grad:
            gl = ctx.meta_out["means2d_grad"]
            m2d.grad = gl[0].unsqueeze(0) if C == 1 else torch.stack(gl)
            if absgrad:
                al = ctx.meta_out["means2d_absgrad"]
                m2d.absgrad = al[0].unsqueeze(0) if C == 1 else torch.stack(al)
        v_bg = None
        if backgrounds is not None and ctx.needs_input_grad[7]:
            if ctx.expected_depth:
                v_render = torch.cat([v_render[..., :-1],
                                      (v_render[..., -1] / alphas.clamp(min=1e-10)).unsqueeze(-1)], dim=-1)
            v_bg = (v_render * (1.0 - alphas).unsqueeze(-1)).sum(dim=(1, 2))
        return (v_means, v_quats, v_scales, v_opacities, v_sh, v_viewmats, None, v_bg) + (None,) * 18


def rasterization(means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor,
                  colors: Tensor, viewmats: Tensor, Ks: Tensor, width: int, height: int,
                  near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0,
                  eps2d: float = 0.3, sh_degree: Optional[int] = None, packed: bool = False,
                  tile_size: int = TILE_SIZE, backgrounds: Optional[Tensor] = None,
                  render_mode: str = "RGB", sparse_grad: bool = False, absgrad: bool = False,
                  rasterize_mode: str = "classic", channel_chunk: int = 32,
                  isect_capacity: Optional[int] = None,
                  tile_bounds: str = "tight",
                  raster_schedule: str = "latency",
                  lean_meta: bool = False,
                  backward_segment: int = 256,
                  radius_rule: str = "classic",
                  dataset_out=None) -> Tuple[Tensor, Tensor, Dict]:
    """Render N Gaussians from C cameras.

    dataset_out = (rgba uint8 [C,H,W,4], distance [C,H,W,1] float16 / 32 / 64 or None, K [3,3], keep_float_frame):
    inference frames ("RGB+ED", lean_meta=True, isect_capacity given) leave the raster as the dataset frames the
    reference's readers open (dataset.frame_to_dataset's bytes) in the caller's buffers; with keep_float_frame False the
    float frame is not written at all and the returned colours / alphas are unspecified.

    radius_rule: "classic" -- gsplat 1.4's single radius ceil(3 sqrt(lambda_1)) per Gaussian (the
    semantics this build's parity claim is made for) -- or "opacity_aware" -- gsplat >= 1.5's per-axis extents
    min(3.33, sqrt(2 ln(255 opacity))) sqrt(Sigma_ii): meta["radii"] is then [C,N,2], Gaussians of
    opacity < 1/255 are culled, n_isect shrinks and pixels change in the corners of the classic square and beyond
    3 sigma of opaque Gaussians.  A compile-time policy of the projection kernels (both instantiations ship).

    backward_segment (SH path, when gradients are wanted): list entries per unit of work of the backward raster
    (a power of two >= 64; 0 = one unit per tile, the whole-list walk).  The training forward stores per-pixel
    checkpoints at that interval ((1 + channels) * 1 KB per tile and segment) and the backward walks the segments
    independently: no tile is one wave's serial job any more.  Gradients equal the whole-list walk's up to rounding.

    lean_meta (SH path, isect_capacity given, no gradients): all C cameras go through ONE C call
    (mgs_render_frames) whose frames keep only what their own kernels read; meta then holds n_isects and
    isect_status [C] and nothing per Gaussian or per tile, and the projection kernel skips 40 MB of stores per
    1 M Gaussians.  FrameRenderer's default.

    raster_schedule (SH path): "latency" runs the tile raster with one wave per 8x8 block (the launch
    has the GPU to itself: a single frame, a training step; -19 % kernel time), "throughput" with
    one wave per tile (7 % fewer vector instructions: several independent frames in flight, where
    other frames' kernels fill the gaps anyway -- FrameRenderer picks it when frames_in_flight > 1).
    The pixels are identical bit for bit.

    tile_bounds (SH path): "tight" bins each Gaussian only into the tiles it can reach with
    alpha >= 1/255 (mgs_isect_tiles with conics + opacities): same image and gradients bit for
    bit, shorter lists; meta["tiles_per_gauss"] / ["n_isects"] then count those lists.
    "classic" reproduces gsplat's mean +- radius rectangles in the meta outputs as well.

    means [N,3], quats [N,4] (wxyz), scales [N,3], opacities [N] (post-activation);
    colors [N,K,3] SH coefficients when sh_degree is given, else [N,D] / [C,N,D] features;
    viewmats [C,4,4] OpenCV world-to-camera; Ks [C,3,3].
    Returns render_colors [C,H,W,D'], render_alphas [C,H,W,1], meta.
    """
    if render_mode not in _MODES:
        raise ValueError(f"render_mode {render_mode!r} not in {_MODES}")
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"rasterize_mode {rasterize_mode!r}")
    if packed or sparse_grad:
        raise NotImplementedError("packed / sparse_grad are not supported")
    if tile_size != TILE_SIZE:
        raise NotImplementedError(f"tile_size must be {TILE_SIZE}")
    if tile_bounds not in ("tight", "classic"):
        raise ValueError(f"tile_bounds {tile_bounds!r} not in ('tight', 'classic')")
    rule = ops.radius_rule_id(radius_rule)
    if backward_segment and (backward_segment < 64 or backward_segment & (backward_segment - 1)):
        raise ValueError(f"backward_segment {backward_segment} is not 0 or a power of two >= 64")
    if raster_schedule not in ("latency", "throughput"):
        raise ValueError(f"raster_schedule {raster_schedule!r} not in ('latency', 'throughput')")
    require_device(means, quats, scales, opacities, colors, viewmats, Ks, backgrounds)
    N, C = means.shape[0], viewmats.shape[0]
    if means.shape != (N, 3) or quats.shape != (N, 4) or scales.shape != (N, 3) \
            or opacities.shape != (N,):
        raise ValueError("expected means [N,3], quats [N,4], scales [N,3], opacities [N]")
    if viewmats.shape != (C, 4, 4) or Ks.shape != (C, 3, 3):
        raise ValueError("expected viewmats [C,4,4], Ks [C,3,3]")
    means, quats, scales, opacities = _f32c(means), _f32c(quats), _f32c(scales), _f32c(opacities)
    colors, viewmats, Ks, backgrounds = _f32c(colors), _f32c(viewmats), _f32c(Ks), _f32c(backgrounds)
    width, height = int(width), int(height)
    antialiased = rasterize_mode == "antialiased"
    want_rgb = render_mode.startswith("RGB")
    want_depth = render_mode != "RGB"
    tile_w, tile_h = -(-width // TILE_SIZE), -(-height // TILE_SIZE)
    meta: Dict = _Meta({"width": width, "height": height, "tile_size": TILE_SIZE,
                        "tile_width": tile_w, "tile_height": tile_h, "n_cameras": C,
                        "camera_ids": None, "gaussian_ids": None})

    # "D" / "ED" of an SH-coloured scene with a fixed list capacity (FrameRenderer, HIP graphs): the fused,
    # read-back-free path renders RGB + depth and the depth channel is sliced off below
    depth_only_via_sh = (sh_degree is not None and not want_rgb and isect_capacity is not None
                         and colors.dim() == 3 and colors.shape[-1] == 3)
    if depth_only_via_sh and backgrounds is not None:
        if backgrounds.shape != (C, 1):
            raise ValueError("backgrounds must be [C, channels]")
        backgrounds = torch.cat([backgrounds.new_zeros(C, 3), backgrounds], dim=-1)
    if sh_degree is not None and (want_rgb or depth_only_via_sh):
        if colors.dim() != 3 or colors.shape[0] != N or colors.shape[2] != 3:
            raise ValueError("with sh_degree set, colors must be [N,K,3]")
        if not 0 <= sh_degree <= 3 or colors.shape[1] < (sh_degree + 1) ** 2:
            raise ValueError("sh_degree outside 0..3 or too few coefficients")
        if backgrounds is not None and backgrounds.shape != (C, 4 if want_depth else 3):
            raise ValueError("backgrounds must be [C, channels]")
        store = meta        # the autograd function publishes per-camera intermediates (and, after
        #                     backward, "means2d_grad" / "means2d_absgrad" lists) into the meta dict
        render, alphas = _RenderSH.apply(
            means, quats, scales, opacities, colors, viewmats, Ks, backgrounds, width, height,
            int(sh_degree), float(eps2d), float(near_plane), float(far_plane),
            float(radius_clip), antialiased, want_depth, isect_capacity, bool(absgrad), store,
            int(tile_bounds == "tight") | (2 if rule else 0), render_mode in ("RGB+ED", "ED"), raster_schedule == "latency",
            bool(lean_meta), int(backward_segment), dataset_out)
        if depth_only_via_sh:
            render = render[..., 3:4]
        if "lean" in store:             # inference frames through mgs_render_frames: counts and status only
            meta.update(store.pop("lean"))
            return render, alphas, meta
        per_cam = store.pop("per_cam")

        def _stk(xs):                 # no copy for the common single-camera call
            return xs[0].unsqueeze(0) if len(xs) == 1 else torch.stack(xs)

        def _cat(xs):
            return xs[0] if len(xs) == 1 else torch.cat(xs)
        if per_cam[0][0] is not None:        # (a lean frame has none of these)
            meta.update(
                radii=_stk([ops.radii_meta(p[0]) for p in per_cam]),
                means2d=_stk([p[1] for p in per_cam]),
                conics=_stk([p[3] for p in per_cam]),
                tiles_per_gauss=_stk([p[6].tiles_per_gauss for p in per_cam]))
        meta.update(
            depths=_stk([p[2] for p in per_cam]),
            opacities=(_stk([p[4] for p in per_cam]) if antialiased
                       else opacities.unsqueeze(0).expand(C, N)),
            n_isects=_cat([p[6].n_isect for p in per_cam]),
            isect_status=_cat([p[6].status for p in per_cam]),
            isect_offsets=_stk([p[6].tile_offsets[:-1].view(tile_h, tile_w) for p in per_cam]),
            tile_lists=[p[6] for p in per_cam])
        if torch.is_grad_enabled() and render.requires_grad and "means2d" in meta:
            meta["means2d"] = meta["means2d"].detach().requires_grad_(True)   # see _RenderSH.backward
    else:
        # feature path: colours are given per Gaussian (or evaluated from SH for "D"/"ED")
        if isect_capacity is not None:
            # this path sizes its lists by reading the intersection count back (as the reference
            # operator does), so it can neither honour a fixed capacity nor be captured in a graph
            raise NotImplementedError(
                "isect_capacity (read-back-free, graph-capturable frames) needs the SH colour path "
                "(sh_degree given with [N,K,3] coefficients); drop isect_capacity for per-Gaussian features")
        radii, means2d, depths, conics, comps = ops.fully_fused_projection(
            means, None, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
            radius_clip, calc_compensations=antialiased, opacities=opacities if rule else None,
            radius_rule=radius_rule)
        opac = opacities.unsqueeze(0).expand(C, N)
        if antialiased:
            opac = opac * comps
        feats = None
        if want_rgb:
            feats = colors.unsqueeze(0).expand(C, N, -1) if colors.dim() == 2 else colors
            if feats.shape[:2] != (C, N):
                raise ValueError("colors must be [N,D] or [C,N,D] when sh_degree is None")
        if want_depth:
            d = depths.unsqueeze(-1)
            feats = d if feats is None else torch.cat([feats, d], dim=-1)
        tpg, isect_ids, flatten_ids = ops.isect_tiles(means2d, radii, depths, TILE_SIZE, tile_w,
                                                      tile_h)
        offsets = ops.isect_offset_encode(isect_ids, C, tile_w, tile_h)
        render, alphas = ops.rasterize_to_pixels(means2d, conics, feats.contiguous(),
                                                 opac.contiguous(), width, height, TILE_SIZE,
                                                 offsets, flatten_ids, backgrounds=backgrounds,
                                                 absgrad=absgrad)
        meta.update(radii=radii, means2d=means2d, depths=depths, conics=conics, opacities=opac,
                    tiles_per_gauss=tpg, isect_ids=isect_ids, flatten_ids=flatten_ids,
                    isect_offsets=offsets)
        if render_mode in ("ED", "RGB+ED"):      # operator path (explicit features): divide here
            render = torch.cat([render[..., :-1],
                                render[..., -1:] / alphas.clamp(min=1e-10)], dim=-1)
    return render, alphas, meta


def check_isect_status(meta: Dict) -> None:
    """Raise if any camera's intersection list overflowed its capacity (reads one word back)."""
    if "isect_status" in meta and bool((meta["isect_status"] != 0).any().item()):
        need = int(meta["n_isects"].max().item())
        raise _lib.MgsError(f"tile-intersection capacity exceeded: a camera needs {need} slots; "
                            "re-render with a larger isect_capacity")


def render(gaussians, cameras: Sequence, sh_degree: Optional[int] = None,
           render_mode: str = "RGB+ED", background: Optional[Sequence[float]] = None,
           device: str = "cuda", tensors: Optional[Dict] = None, **kw):
    """Render a `Gaussians` scene from `Camera`s (the Python-side API that stays, per the
    north star).  Applies splatfacto's post-processing: rgb = clamp(rgb + (1-alpha)*bg, 0, 1),
    depth = where(alpha > 0, depth, max depth).  Returns dict(rgb, depth, alpha, meta)."""
    cams = list(cameras)
    w, h = cams[0].width, cams[0].height
    if any(c.width != w or c.height != h for c in cams):
        raise ValueError("all cameras of one call must share a resolution")
    t = tensors if tensors is not None else gaussians.to_torch(device, sh_degree)
    viewmats = torch.from_numpy(np.stack([c.viewmat() for c in cams]).astype(np.float32)).to(device)
    Ks = torch.from_numpy(np.stack([c.K for c in cams]).astype(np.float32)).to(device)
    colors, alphas, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"],
                                         t["colors"], viewmats, Ks, w, h,
                                         near_plane=cams[0].near, far_plane=cams[0].far,
                                         sh_degree=t["sh_degree"], render_mode=render_mode, **kw)
    out = {"alpha": alphas, "meta": meta}
    if render_mode.startswith("RGB"):
        rgb = colors[..., :3]
        if background is not None:
            bg = torch.tensor(background, dtype=torch.float32, device=rgb.device)
            rgb = rgb + (1.0 - alphas) * bg
        out["rgb"] = rgb.clamp(0.0, 1.0)
    if render_mode != "RGB":
        depth = colors[..., -1:]
        out["depth"] = torch.where(alphas > 0, depth, depth.detach().max())
    return out
