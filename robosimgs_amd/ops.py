"""Stage operators of the render path, gsplat-1.x-compatible signatures (SURVEY.md 8(b),
Appendix A.1), each a thin torch wrapper over one C-ABI entry point of libmgs.so.

    fully_fused_projection   -> mgs_projection_fwd / mgs_projection_bwd
    spherical_harmonics      -> mgs_sh_fwd / mgs_sh_bwd
    isect_tiles              -> mgs_isect_tiles
    isect_offset_encode      -> mgs_isect_offset_encode
    rasterize_to_pixels      -> mgs_rasterize_fwd / mgs_rasterize_bwd

Inputs are post-activation fp32 CUDA(HIP) tensors; ids are int32, keys int64.  All kernels
are enqueued on torch's current stream.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check, ptr, require_device, stream_handle

TILE_SIZE = 16


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ======================================================================================
# single-camera raw calls (no autograd); used by the operators below and by rendering.py
# ======================================================================================
RADIUS_RULES = {"classic": 0, "opacity_aware": 1}     # include/mgs.h MGS_RADIUS_CLASSIC / MGS_RADIUS_OPACITY_AWARE


def radius_rule_id(radius_rule) -> int:
    if radius_rule not in RADIUS_RULES:
        raise ValueError(f"radius_rule {radius_rule!r} not in {tuple(RADIUS_RULES)}")
    return RADIUS_RULES[radius_rule]


def radii_x(radii):
    """The per-Gaussian visibility / x-extent array of either radii layout: [N] (classic rule) or the planar [2,N]
    pair of per-axis extents the raw calls keep under the opacity-aware rule."""
    return radii if radii is None or radii.dim() == 1 else radii[0]


def radii_meta(radii):
    """Radii as the operator returns them: [N] (gsplat 1.4) or [N,2] (gsplat >= 1.5, per axis)."""
    return radii if radii is None or radii.dim() == 1 else radii.t()


def projection_fwd_raw(means, quats, scales, viewmat, K, width, height, eps2d, near_plane,
                       far_plane, radius_clip, calc_compensations, opacities=None, radius_rule=0):
    """radius_rule 1 (MGS_RADIUS_OPACITY_AWARE): radii comes back planar [2,N] (x extents, y extents)."""
    n = means.shape[0]
    dev = means.device
    radii = torch.empty((2, n) if radius_rule else (n,), dtype=torch.int32, device=dev)
    means2d = torch.empty(n, 2, dtype=torch.float32, device=dev)
    depths = torch.empty(n, dtype=torch.float32, device=dev)
    conics = torch.empty(n, 3, dtype=torch.float32, device=dev)
    comp = torch.empty(n, dtype=torch.float32, device=dev) if calc_compensations else None
    check(_lib.lib().mgs_projection_fwd(n, ptr(means), ptr(quats), ptr(scales), ptr(viewmat),
                                        ptr(K), width, height, eps2d, near_plane, far_plane,
                                        radius_clip, ptr(radii_x(radii)), ptr(means2d), ptr(depths),
                                        ptr(conics), ptr(comp), ptr(opacities) if radius_rule else None,
                                        int(radius_rule), ptr(radii[1]) if radius_rule else None, stream_handle()),
          "mgs_projection_fwd")
    return radii, means2d, depths, conics, comp


def project_color_fwd_raw(means, quats, scales, opacities, sh_degree, sh_coeffs, viewmat, K,
                          width, height, eps2d, near_plane, far_plane, radius_clip,
                          antialiased, with_depth, want_splats=False, bin_seed=None, lean=False, per_axis=False):
    """Returns (radii, means2d, depths, conics, opac_aa|None, feats) and, with want_splats, a 7th
    item: the packed [N,12] records the raster kernels gather from.  bin_seed = "tight" | "classic":
    an 8th item (seed_info [N,2] i32, seed_sums [ceil(N/64)] i32) for isect_tiles_raw(seed=...).
    lean (needs want_splats and bin_seed): radii / means2d / conics / feats are not written and come back
    as None -- an inference frame, whose raster reads the records and whose binning reads the seed.
    per_axis: project with MGS_RADIUS_OPACITY_AWARE; radii is then planar [2,N] (radii_x / radii_meta)."""
    n = means.shape[0]
    dev = means.device
    if lean and not (want_splats and bin_seed is not None):
        raise ValueError("lean=True needs want_splats=True and a bin_seed")
    stride = 4 if with_depth else 3
    radii = means2d = conics = feats = None
    if not lean:
        radii = torch.empty((2, n) if per_axis else (n,), dtype=torch.int32, device=dev)
        means2d = torch.empty(n, 2, dtype=torch.float32, device=dev)
        conics = torch.empty(n, 3, dtype=torch.float32, device=dev)
        feats = torch.empty(n, stride, dtype=torch.float32, device=dev)
    depths = torch.empty(n, dtype=torch.float32, device=dev)
    opac = torch.empty(n, dtype=torch.float32, device=dev) if antialiased else None
    splats = torch.empty(n, 12, dtype=torch.float32, device=dev) if want_splats else None
    if splats is not None:
        _splat_annotation.pop(splats.data_ptr(), None)     # fresh records: no binning's slot words in them
    seed = None
    if bin_seed is not None and n > 0:
        seed = (torch.empty(n, 2, dtype=torch.int32, device=dev),
                torch.empty(max(1, -(-n // 64)), dtype=torch.int32, device=dev))
    check(_lib.lib().mgs_project_color_fwd(
        n, ptr(means), ptr(quats), ptr(scales), ptr(opacities), sh_degree, sh_coeffs.shape[1],
        ptr(sh_coeffs), ptr(viewmat), ptr(K), width, height, eps2d, near_plane, far_plane,
        radius_clip, ptr(radii_x(radii)), ptr(means2d), ptr(depths), ptr(conics), ptr(opac), stride,
        ptr(feats), ptr(splats), int(bin_seed == "tight") | (2 if per_axis else 0), ptr(seed[0]) if seed else None,
        ptr(seed[1]) if seed else None, ptr(radii[1]) if (per_axis and radii is not None) else None, stream_handle()),
        "mgs_project_color_fwd")
    out = (radii, means2d, depths, conics, opac, feats)
    if want_splats or bin_seed is not None:
        out = out + (splats,)
    if bin_seed is not None:
        out = out + (seed,)
    return out


class TileLists:
    """Depth-ordered per-tile lists of one camera (device resident, capacity sized)."""
    __slots__ = ("n_isect", "tile_ids", "flatten_ids", "tile_offsets", "tiles_per_gauss",
                 "isect_ids", "status", "capacity", "pair_info", "group_order", "splat_slots")


_workspaces: dict = {}

# mgs_isect_tiles(splat_slots=) writes a binning's record slots into words 10-11 of the caller's splat records IN PLACE:
# the records then belong to THAT binning.  data_ptr of an annotated record tensor -> the generation of the binning that
# annotated it last; a TileLists remembers (data_ptr, generation) and rasterize_bwd_det_raw trusts the records' slot words
# only while the pair still matches (otherwise it gathers pair_info, which every TileLists owns).
_splat_annotation: dict = {}
_splat_generation = [0]


def _workspace(nbytes: int, device) -> Tensor:
    """One growing scratch tensor per (device, stream).  Stream-ordered reuse is safe because
    every consumer is enqueued on the same stream."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream,
           torch.cuda.is_current_stream_capturing())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def isect_tiles_raw(means2d, radii, depths, tile_w, tile_h, capacity: int, cam_id=0, n_cams=1,
                    want_isect_ids=False, want_tiles_per_gauss=True,
                    want_pair_info=False, conics=None, opacities=None, seed=None,
                    want_tile_ids=True, want_group_order=True, splats=None) -> TileLists:
    """conics + opacities given: tile rectangles tightened to the tiles a Gaussian can reach with
    alpha >= 1/255 (shorter lists, bit-identical render); None: gsplat's classic rectangles.
    seed = (seed_info, seed_sums) from project_color_fwd_raw(bin_seed=...): the rectangles come from
    there (means2d / radii / conics / opacities are then not read and may be None; seed_sums is consumed).
    radii: [N], or planar [2,N] per-axis extents (gsplat >= 1.5's rule).
    splats (with want_pair_info): the packed records [N,12] of project_color_fwd_raw; the pairs' record slots are left in
    their padding words IN PLACE (include/mgs.h: splat_slots) -- the records are tied to this binning from then on -- and the
    lists remember which tensor they annotated (`splat_slots`): rasterize_bwd_det_raw gathers nothing but the record while
    that pairing holds, and falls back to pair_info when the records were re-projected or binned again since."""
    n = depths.shape[0]
    dev = depths.device
    L = _lib.lib()
    out = TileLists()
    out.capacity = int(capacity)
    out.n_isect = torch.empty(1, dtype=torch.int32, device=dev)
    out.status = torch.empty(1, dtype=torch.int32, device=dev)       # written by every call
    out.tile_ids = torch.empty(capacity, dtype=torch.int32, device=dev) if want_tile_ids else None
    out.flatten_ids = torch.empty(capacity, dtype=torch.int32, device=dev)
    out.tile_offsets = torch.empty(tile_w * tile_h + 1, dtype=torch.int32, device=dev)
    out.tiles_per_gauss = (torch.empty(n, dtype=torch.int32, device=dev)
                           if want_tiles_per_gauss else None)
    out.isect_ids = torch.empty(capacity, dtype=torch.int64, device=dev) if want_isect_ids else None
    out.pair_info = torch.empty(n, 4, dtype=torch.int32, device=dev) if want_pair_info else None
    # launch order of the raster kernels' tiles (groups of four, longest lists first): a schedule, not a result
    out.group_order = (torch.empty((tile_w * tile_h + 3) // 4, dtype=torch.int32, device=dev) if want_group_order else None)
    nbytes = ctypes.c_size_t(0)
    args = [n, ptr(means2d), ptr(radii_x(radii)), ptr(radii[1]) if (radii is not None and radii.dim() == 2) else None,
            ptr(depths), ptr(conics), ptr(opacities), TILE_SIZE,
            tile_w, tile_h, cam_id, n_cams,
            capacity, ptr(out.tiles_per_gauss), ptr(out.n_isect), ptr(out.tile_ids),
            ptr(out.flatten_ids), ptr(out.isect_ids), ptr(out.tile_offsets), ptr(out.pair_info),
            ptr(out.group_order), ptr(out.status), ptr(seed[0]) if seed else None, ptr(seed[1]) if seed else None,
            ptr(splats) if (splats is not None and want_pair_info) else None]
    out.splat_slots = False
    if splats is not None and want_pair_info:
        _splat_generation[0] += 1
        _splat_annotation[splats.data_ptr()] = _splat_generation[0]
        out.splat_slots = (splats.data_ptr(), _splat_generation[0])
    check(L.mgs_isect_tiles(*args, None, ctypes.byref(nbytes), stream_handle()),
          "mgs_isect_tiles(size query)")
    ws = _workspace(nbytes.value, dev)
    nbytes = ctypes.c_size_t(ws.numel())
    check(L.mgs_isect_tiles(*args, ptr(ws), ctypes.byref(nbytes), stream_handle()),
          "mgs_isect_tiles")
    return out


def render_frames_raw(means, quats, scales, opacities, sh_degree, sh_coeffs, viewmats, Ks, width, height,
                      eps2d, near_plane, far_plane, radius_clip, antialiased, with_depth, capacity,
                      backgrounds=None, expected_last=False, latency=False, out=None, tight=True, per_axis=False,
                      dataset=None, float_frame=True):
    """mgs_render_frames: C inference frames in one C call (no per-Gaussian outputs, scratch reused from camera to
    camera).  viewmats [C,4,4], Ks [C,3,3], backgrounds [C,ch] or None.  Returns (render [C,H,W,ch], alphas [C,H,W],
    n_isects [C] i32, isect_status [C] i32); out = (render, alphas) to write into existing buffers.
    tight=False: gsplat's classic tile rectangles (MGS_FRAMES_CLASSIC_BOUNDS; same pixels, classic counts).
    dataset = (rgba uint8 [C,H,W,4], distance [C,H,W,1] or None, K): the dataset frames straight out of the raster
    (with_depth and expected_last required); float_frame=False then leaves render / alphas unwritten."""
    dev = means.device
    C, n = viewmats.shape[0], means.shape[0]
    ch = 4 if with_depth else 3
    if out is None:
        render = torch.empty(C, height, width, ch, dtype=torch.float32, device=dev)
        alphas = torch.empty(C, height, width, dtype=torch.float32, device=dev)
    else:
        render, alphas = out
    n_isect = torch.empty(C, dtype=torch.int32, device=dev)
    status = torch.empty(C, dtype=torch.int32, device=dev)
    L = _lib.lib()
    nbytes = ctypes.c_size_t(0)
    args = [n, ptr(means), ptr(quats), ptr(scales), ptr(opacities), int(sh_degree), sh_coeffs.shape[1], ptr(sh_coeffs),
            C, ptr(viewmats), ptr(Ks), int(width), int(height), eps2d, near_plane, far_plane, radius_clip,
            int(bool(antialiased)), ch,
            int(bool(expected_last)) | (2 if latency else 0) | (0 if tight else 4) | (8 if per_axis else 0),
            ptr(backgrounds),
            int(capacity), ptr(render) if (float_frame or dataset is None) else None,
            ptr(alphas) if (float_frame or dataset is None) else None, ptr(n_isect), ptr(status)]
    ds = dataset_args(dataset, C, height, width, dev)
    args += list(ds[:4])
    check(L.mgs_render_frames(*args, None, ctypes.byref(nbytes), stream_handle()), "mgs_render_frames(size query)")
    ws = _workspace(nbytes.value + 256, dev)
    base = ws.data_ptr()
    aligned = (base + 255) // 256 * 256
    nbytes = ctypes.c_size_t(ws.numel() - (aligned - base))
    check(L.mgs_render_frames(*args, aligned, ctypes.byref(nbytes), stream_handle()), "mgs_render_frames")
    return render, alphas, n_isect, status


def checkpoint_buffer(capacity: int, tile_w: int, tile_h: int, channels: int, interval: int, device) -> Tensor:
    """The buffer rasterize_fwd_raw(checkpoints=...) writes for lists of up to `capacity` entries."""
    n = _lib.lib().mgs_raster_checkpoint_floats(int(capacity), tile_w, tile_h, channels, int(interval))
    if n == 0:
        raise ValueError(f"checkpoint interval {interval} is not a power of two >= 64")
    return torch.empty(n, dtype=torch.float32, device=device)


TRAIN_FIELDS = ("radii", "means2d", "depths", "conics", "opac_aa", "feats", "splats", "tiles_per_gauss", "pair_info",
                "tile_ids", "flatten_ids", "tile_offsets", "group_order", "last_ids", "checkpoints", "counts", "radii_y")


class TrainState:
    """Per-camera state of a batch of training frames (mgs_render_frames_train writes it, mgs_render_frames_backward
    reads it): one device buffer, fields at the offsets mgs_train_state_layout reports; `views(c)` hands them out as
    tensors without copying."""

    def __init__(self, n, n_cams, width, height, channels, capacity, antialiased, interval, device):
        offs = (ctypes.c_size_t * len(TRAIN_FIELDS))()
        per = ctypes.c_size_t(0)
        check(_lib.lib().mgs_train_state_layout(n, width, height, channels, int(capacity), int(bool(antialiased)), int(interval),
                                                offs, ctypes.byref(per)), "mgs_train_state_layout")
        self.offsets, self.per_camera = list(offs), per.value
        self.n, self.n_cams, self.width, self.height, self.channels = n, n_cams, width, height, channels
        self.capacity, self.antialiased, self.interval = int(capacity), bool(antialiased), int(interval)
        self.buf = torch.empty(self.per_camera * n_cams + 256, dtype=torch.uint8, device=device)
        self.pad = (-self.buf.data_ptr()) % 256
        self.n_tiles = (-(-width // TILE_SIZE)) * (-(-height // TILE_SIZE))

    def ptr(self) -> int:
        return self.buf.data_ptr() + self.pad

    def views(self, c: int) -> dict:
        n, cap, nt = self.n, self.capacity, self.n_tiles
        shapes = {"radii": (torch.int32, (n,)), "means2d": (torch.float32, (n, 2)), "depths": (torch.float32, (n,)),
                  "conics": (torch.float32, (n, 3)), "opac_aa": (torch.float32, (n,) if self.antialiased else (0,)),
                  "feats": (torch.float32, (n, self.channels)), "splats": (torch.float32, (n, 12)),
                  "tiles_per_gauss": (torch.int32, (n,)), "pair_info": (torch.int32, (n, 4)), "tile_ids": (torch.int32, (cap,)),
                  "flatten_ids": (torch.int32, (cap,)), "tile_offsets": (torch.int32, (nt + 1,)),
                  "group_order": (torch.int32, ((nt + 3) // 4,)), "last_ids": (torch.int32, (self.height, self.width)),
                  "counts": (torch.int32, (2,)), "radii_y": (torch.int32, (n,))}
        out = {}
        base = self.pad + self.per_camera * c
        for name, off in zip(TRAIN_FIELDS, self.offsets):
            if name not in shapes:
                continue
            dt, shape = shapes[name]
            numel = 1
            for d_ in shape:
                numel *= d_
            nbytes = numel * (4)
            out[name] = self.buf[base + off:base + off + nbytes].view(dt).view(shape)
        return out

    def tile_lists(self, c: int, v: dict = None) -> "TileLists":
        v = v or self.views(c)
        tl = TileLists()
        tl.capacity = self.capacity
        tl.n_isect, tl.status = v["counts"][0:1], v["counts"][1:2]
        tl.tile_ids, tl.flatten_ids, tl.tile_offsets = v["tile_ids"], v["flatten_ids"], v["tile_offsets"]
        tl.tiles_per_gauss, tl.pair_info, tl.group_order, tl.isect_ids = v["tiles_per_gauss"], v["pair_info"], v["group_order"], None
        tl.splat_slots = True              # mgs_render_frames_train annotates the state's splat records
        return tl


def _aligned_ws(nbytes, dev):
    ws = _workspace(nbytes + 256, dev)
    base = ws.data_ptr()
    aligned = (base + 255) // 256 * 256
    return aligned, ws.numel() - (aligned - base)


def render_frames_train_raw(means, quats, scales, opacities, sh_degree, sh_coeffs, viewmats, Ks, width, height, eps2d,
                            near_plane, far_plane, radius_clip, antialiased, with_depth, capacity, interval,
                            backgrounds=None, expected_last=False, latency=True, tight=True, out=None, per_axis=False):
    """mgs_render_frames_train: C training frames in one C call.  Returns (render [C,H,W,ch], alphas [C,H,W], TrainState)."""
    dev = means.device
    C, n = viewmats.shape[0], means.shape[0]
    ch = 4 if with_depth else 3
    if out is None:
        render = torch.empty(C, height, width, ch, dtype=torch.float32, device=dev)
        alphas = torch.empty(C, height, width, dtype=torch.float32, device=dev)
    else:
        render, alphas = out
    st = TrainState(n, C, width, height, ch, capacity, antialiased, interval, dev)
    flags = int(bool(expected_last)) | (2 if latency else 0) | (0 if tight else 4) | (8 if per_axis else 0)
    L = _lib.lib()
    nbytes = ctypes.c_size_t(0)
    args = [n, ptr(means), ptr(quats), ptr(scales), ptr(opacities), int(sh_degree), sh_coeffs.shape[1], ptr(sh_coeffs), C,
            ptr(viewmats), ptr(Ks), int(width), int(height), eps2d, near_plane, far_plane, radius_clip, int(bool(antialiased)),
            ch, flags, ptr(backgrounds), int(capacity), int(interval), ptr(render), ptr(alphas), st.ptr()]
    check(L.mgs_render_frames_train(*args, None, ctypes.byref(nbytes), stream_handle()), "mgs_render_frames_train(size query)")
    aligned, room = _aligned_ws(nbytes.value, dev)
    nbytes = ctypes.c_size_t(room)
    check(L.mgs_render_frames_train(*args, aligned, ctypes.byref(nbytes), stream_handle()), "mgs_render_frames_train")
    st.flags = flags
    return render, alphas, st


def render_frames_backward_raw(means, quats, scales, opacities, sh_degree, sh_coeffs, viewmats, Ks, eps2d, backgrounds, st,
                               render, alphas, v_render, v_alphas, absgrad=False, want_viewmats=False):
    """mgs_render_frames_backward on a TrainState.  Returns (v_means, v_quats, v_scales, v_sh, v_opacities, v_viewmats|None,
    v_means2d [C,N,2], v_means2d_abs [C,N,2]|None)."""
    dev = means.device
    C, n = viewmats.shape[0], means.shape[0]
    v_means, v_quats, v_scales = torch.empty_like(means), torch.empty_like(quats), torch.empty_like(scales)
    v_sh, v_opac = torch.empty_like(sh_coeffs), torch.empty_like(opacities)
    v_vm = torch.zeros_like(viewmats) if want_viewmats else None
    v_m2d = torch.empty(C, n, 2, dtype=torch.float32, device=dev)
    v_abs = torch.empty(C, n, 2, dtype=torch.float32, device=dev) if absgrad else None
    L = _lib.lib()
    nbytes = ctypes.c_size_t(0)
    args = [n, ptr(means), ptr(quats), ptr(scales), ptr(opacities), int(sh_degree), sh_coeffs.shape[1], ptr(sh_coeffs), C,
            ptr(viewmats), ptr(Ks), st.width, st.height, eps2d, int(st.antialiased), st.channels, st.flags, ptr(backgrounds),
            st.capacity, st.interval, ptr(render), ptr(alphas), ptr(v_render), ptr(v_alphas), st.ptr(), ptr(v_means),
            ptr(v_quats), ptr(v_scales), ptr(v_sh), ptr(v_opac), ptr(v_vm), ptr(v_m2d), ptr(v_abs)]
    check(L.mgs_render_frames_backward(*args, None, ctypes.byref(nbytes), stream_handle()), "mgs_render_frames_backward(size query)")
    ws = torch.empty(nbytes.value + 256, dtype=torch.uint8, device=dev)
    aligned = (ws.data_ptr() + 255) // 256 * 256
    nbytes = ctypes.c_size_t(ws.numel() - (aligned - ws.data_ptr()))
    check(L.mgs_render_frames_backward(*args, aligned, ctypes.byref(nbytes), stream_handle()), "mgs_render_frames_backward")
    return v_means, v_quats, v_scales, v_sh, v_opac, v_vm, v_m2d, v_abs


def dataset_args(dataset, n_frames=None, height=None, width=None, device=None):
    """(ds_rgba, ds_distance, ds_distance_type, ds_Kinv_host, keep-alive) for the C calls that take a dataset output;
    dataset = (rgba uint8 [..,H,W,4], distance [..,H,W,1] or None, K [3,3]) or None.  The kernels write
    n_frames * height * width * 4 bytes of RGBA and as many distances through raw pointers: the buffers must hold exactly
    that, on the device the frame is rendered on (checked here; one K serves every camera of the call -- the
    intrinsics a dataset's cameras share)."""
    if dataset is None:
        return None, None, 0, None, None
    import numpy as np
    rgba, dist, K = dataset
    if rgba.dtype != torch.uint8 or not rgba.is_contiguous():
        raise ValueError("dataset rgba must be a contiguous uint8 tensor [..., H, W, 4]")
    if dist is not None and (not dist.is_contiguous() or dist.dtype not in (torch.float16, torch.float32, torch.float64)):
        raise ValueError("dataset distance must be a contiguous float16 / float32 / float64 tensor [..., H, W, 1]")
    require_device(rgba, dist)
    if device is not None and any(x is not None and x.device != device for x in (rgba, dist)):
        raise ValueError(f"dataset buffers must live on {device}, the device the frame is rendered on")
    if n_frames is not None:
        n_px = int(n_frames) * int(height) * int(width)
        if rgba.numel() != n_px * 4:
            raise ValueError(f"dataset rgba holds {rgba.numel()} bytes, {n_frames} frame(s) of {width}x{height} need {n_px * 4} "
                             f"([{n_frames},{height},{width},4] uint8)")
        if dist is not None and dist.numel() != n_px:
            raise ValueError(f"dataset distance holds {dist.numel()} values, {n_frames} frame(s) of {width}x{height} need {n_px} "
                             f"([{n_frames},{height},{width},1])")
    if dist is not None and K is None:
        raise ValueError("a dataset distance plane needs K (the 3x3 intrinsics the cameras of the call share)")
    kinv = np.ascontiguousarray(np.linalg.inv(np.asarray(K, dtype=np.float64).reshape(3, 3))) if dist is not None else None
    dtype_id = {torch.float64: 1, torch.float16: 2}.get(dist.dtype, 0) if dist is not None else 0
    return ptr(rgba), ptr(dist), dtype_id, (kinv.ctypes.data if kinv is not None else None), kinv


def rasterize_fwd_raw(means2d, conics, feats, opacities, background, width, height, tile_w,
                      tile_h, tile_offsets, flatten_ids, out=None, track_last=True, splats=None,
                      expected_last=False, latency=False, group_order=None, channels=None,
                      checkpoints=None, checkpoint_interval=0, dataset=None):
    """out = (render, alphas, last_ids|None) to write into existing buffers.
    dataset = (rgba uint8 [H,W,4], distance [H,W,1] f16 / f32 / f64 or None, K 3x3 numpy): the dataset frame straight
    out of the raster (include/mgs.h: ds_* arguments; 4 channels, expected_last, inference); out = (None, None, None)
    then skips the float frame altogether.  With `splats` (<= 4 channels)
    means2d / conics / feats / opacities are not read and may be None; give `channels` then.  track_last=False (or
    last_ids None) is the inference variant: no last_ids, one select less per pair.
    expected_last: the last channel leaves divided by max(alpha, 1e-10) ("ED" modes).
    latency: MGS_RASTER_LATENCY -- one wave per 8x8 block; faster when the launch has the GPU to itself
    (a single frame, a training step), slower in total work when several frames are in flight.
    group_order: TileLists.group_order -- the tiles are then started longest lists first.
    checkpoints (checkpoint_buffer(...)) + checkpoint_interval: the training variant also stores every pixel's state
    every `checkpoint_interval` list entries, for rasterize_bwd_det_raw(checkpoints=...)."""
    src = means2d if means2d is not None else splats
    n = src.shape[0]
    ch = int(channels) if channels is not None else feats.shape[-1]
    dev = src.device
    if out is None:
        render = torch.empty(height, width, ch, dtype=torch.float32, device=dev)
        alphas = torch.empty(height, width, dtype=torch.float32, device=dev)
        last_ids = (torch.empty(height, width, dtype=torch.int32, device=dev) if track_last
                    else None)
    else:
        render, alphas, last_ids = out
    ds = dataset_args(dataset, 1, height, width, dev)
    check(_lib.lib().mgs_rasterize_fwd(n, ptr(means2d), ptr(conics), ptr(feats), ptr(opacities),
                                       ptr(splats), ptr(background), ch, width, height, tile_w, tile_h,
                                       ptr(tile_offsets), ptr(flatten_ids), ptr(group_order),
                                       int(bool(expected_last)) | (2 if latency else 0),
                                       ptr(render), ptr(alphas), ptr(last_ids), ptr(checkpoints),
                                       int(checkpoint_interval), *ds[:4], stream_handle()),
          "mgs_rasterize_fwd")
    return render, alphas, last_ids


def rasterize_bwd_raw(means2d, conics, feats, opacities, background, width, height, tile_w,
                      tile_h, tile_offsets, flatten_ids, alphas, last_ids, v_render, v_alphas,
                      absgrad=False, accum=None):
    """Returns (v_means2d, v_conics, v_feats, v_opacities, v_means2d_abs|None).  `accum`
    supplies pre-zeroed / partially accumulated output buffers (float atomics add into them)."""
    n = means2d.shape[0]
    ch = feats.shape[-1]
    dev = means2d.device
    if accum is None:
        v_means2d = torch.zeros(n, 2, dtype=torch.float32, device=dev)
        v_conics = torch.zeros(n, 3, dtype=torch.float32, device=dev)
        v_feats = torch.zeros(n, ch, dtype=torch.float32, device=dev)
        v_opac = torch.zeros(n, dtype=torch.float32, device=dev)
        v_abs = torch.zeros(n, 2, dtype=torch.float32, device=dev) if absgrad else None
    else:
        v_means2d, v_conics, v_feats, v_opac, v_abs = accum
    check(_lib.lib().mgs_rasterize_bwd(
        n, ptr(means2d), ptr(conics), ptr(feats), ptr(opacities), ptr(background), ch, width,
        height, tile_w, tile_h, ptr(tile_offsets), ptr(flatten_ids), ptr(alphas), ptr(last_ids),
        ptr(v_render), ptr(v_alphas), ptr(v_means2d), ptr(v_abs), ptr(v_conics), ptr(v_feats),
        ptr(v_opac), stream_handle()), "mgs_rasterize_bwd")
    return v_means2d, v_conics, v_feats, v_opac, v_abs


def _splat_slots_valid(tl, splats) -> bool:
    """The records `splats` carry the record slots of the binning that made `tl` (see _splat_annotation): True for the
    train-state lists (mgs_render_frames_train annotates its own state), else only when `splats` is the very tensor that
    binning annotated and no later binning (or projection) has rewritten it."""
    tag = getattr(tl, "splat_slots", False)
    if splats is None or not tag:
        return False
    if tag is True:
        return True
    return tag[0] == splats.data_ptr() and _splat_annotation.get(tag[0]) == tag[1]


def rasterize_bwd_det_raw(means2d, conics, feats, opacities, background, width, height, tile_w,
                          tile_h, tl: "TileLists", alphas, last_ids, v_render, v_alphas,
                          absgrad=False, splats=None, canary_bytes=0, expected_render=None,
                          render_out=None, checkpoints=None, checkpoint_interval=0, records_only=False):
    """Atomic-free, bit-reproducible raster backward (needs tl.pair_info from the binning).
    Returns freshly written (v_means2d, v_conics, v_feats, v_opacities, v_means2d_abs|None).
    expected_render: the forward's frame when it ran with expected_last (the kernel undoes the divide).
    checkpoints + checkpoint_interval + render_out (the forward's frame): the segmented walk (include/mgs.h).
    records_only: stop after the raster kernel (MGS_RASTER_BWD_RECORDS_ONLY; the returned tensors are not written).
    canary_bytes (tests): that many 0xA5 bytes are kept behind the workspace the library asked for
    and returned as a sixth value, so a test can see that nothing was written past the workspace."""
    n = means2d.shape[0]
    ch = feats.shape[-1]
    dev = means2d.device
    v_means2d = torch.empty(n, 2, dtype=torch.float32, device=dev)
    v_conics = torch.empty(n, 3, dtype=torch.float32, device=dev)
    v_feats = torch.empty(n, ch, dtype=torch.float32, device=dev)
    v_opac = torch.empty(n, dtype=torch.float32, device=dev)
    v_abs = torch.empty(n, 2, dtype=torch.float32, device=dev) if absgrad else None
    L = _lib.lib()
    nbytes = ctypes.c_size_t(0)
    args = [n, ptr(means2d), ptr(conics), ptr(feats), ptr(opacities), ptr(splats), ptr(background),
            ch, width, height, tile_w, tile_h, ptr(tl.tile_offsets), ptr(tl.flatten_ids), ptr(alphas),
            ptr(last_ids), ptr(v_render), ptr(v_alphas), ptr(expected_render), ptr(tl.pair_info),
            ptr(getattr(tl, "group_order", None)), tl.capacity,
            ptr(render_out), ptr(checkpoints), int(checkpoint_interval),
            int(bool(records_only)) | (2 if _splat_slots_valid(tl, splats) else 0),
            ptr(v_means2d), ptr(v_abs), ptr(v_conics), ptr(v_feats), ptr(v_opac)]
    check(L.mgs_rasterize_bwd_det(*args, None, ctypes.byref(nbytes), stream_handle()),
          "mgs_rasterize_bwd_det(size query)")
    ws = torch.full((nbytes.value + canary_bytes,), 0xA5, dtype=torch.uint8, device=dev) if canary_bytes \
        else torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    check(L.mgs_rasterize_bwd_det(*args, ptr(ws), ctypes.byref(nbytes), stream_handle()),
          "mgs_rasterize_bwd_det")
    if canary_bytes:
        return v_means2d, v_conics, v_feats, v_opac, v_abs, ws[nbytes.value:]
    return v_means2d, v_conics, v_feats, v_opac, v_abs


# ======================================================================================
# gsplat-compatible operators (multi-camera, autograd)
# ======================================================================================
class _Projection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane,
                far_plane, radius_clip, calc_compensations, opacities=None, radius_rule=0):
        C = viewmats.shape[0]
        outs = [projection_fwd_raw(means, quats, scales, viewmats[c], Ks[c], width, height,
                                   eps2d, near_plane, far_plane, radius_clip,
                                   calc_compensations, opacities, radius_rule) for c in range(C)]
        radii_out = torch.stack([radii_meta(o[0]) for o in outs])      # [C,N], or [C,N,2] under the per-axis rule
        radii = torch.stack([radii_x(o[0]) for o in outs])             # [C,N]: > 0 = visible (what the backward reads)
        means2d = torch.stack([o[1] for o in outs])
        depths = torch.stack([o[2] for o in outs])
        conics = torch.stack([o[3] for o in outs])
        comps = torch.stack([o[4] for o in outs]) if calc_compensations else None
        ctx.save_for_backward(means, quats, scales, viewmats, Ks, radii, conics, comps)
        ctx.dims = (width, height, eps2d)
        ctx.mark_non_differentiable(radii_out)
        return radii_out, means2d, depths, conics, comps

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comps):
        means, quats, scales, viewmats, Ks, radii, conics, comps = ctx.saved_tensors
        width, height, eps2d = ctx.dims
        n, C = means.shape[0], viewmats.shape[0]
        v_means = torch.zeros_like(means)
        v_quats = torch.zeros_like(quats)
        v_scales = torch.zeros_like(scales)
        want_view = ctx.needs_input_grad[3]
        v_viewmats = torch.zeros_like(viewmats) if want_view else None
        v_means2d, v_depths, v_conics = _f32c(v_means2d), _f32c(v_depths), _f32c(v_conics)
        v_comps = _f32c(v_comps) if comps is not None else None
        for c in range(C):
            check(_lib.lib().mgs_projection_bwd(
                n, ptr(means), ptr(quats), ptr(scales), ptr(viewmats[c]), ptr(Ks[c]), width,
                height, eps2d, ptr(radii[c]), ptr(conics[c]),
                ptr(comps[c]) if comps is not None else None,
                ptr(v_means2d[c]), ptr(v_depths[c]), ptr(v_conics[c]),
                ptr(v_comps[c]) if v_comps is not None else None, ptr(v_means), ptr(v_quats),
                ptr(v_scales), ptr(v_viewmats[c]) if want_view else None, stream_handle()),
                "mgs_projection_bwd")
        return (v_means, v_quats, v_scales, v_viewmats, None, None, None, None, None, None,
                None, None, None, None)


def fully_fused_projection(means: Tensor, covars: Optional[Tensor], quats: Tensor,
                           scales: Tensor, viewmats: Tensor, Ks: Tensor, width: int,
                           height: int, eps2d: float = 0.3, near_plane: float = 0.01,
                           far_plane: float = 1e10, radius_clip: float = 0.0,
                           packed: bool = False, sparse_grad: bool = False,
                           calc_compensations: bool = False, opacities: Optional[Tensor] = None,
                           radius_rule: str = "classic"
                           ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Optional[Tensor]]:
    """World -> screen EWA projection of N Gaussians for C cameras.
    Returns radii [C,N] i32, means2d [C,N,2], depths [C,N], conics [C,N,3],
    compensations [C,N] | None.
    radius_rule: "classic" (gsplat 1.4, SURVEY.md A.2 step 5: one radius ceil(3 sqrt(lambda_1))) or "opacity_aware"
    (gsplat >= 1.5, SURVEY.md A.4: per-axis extents min(3.33, sqrt(2 ln(255 opacity))) sqrt(Sigma_ii), radii [C,N,2];
    `opacities` [N] optional as in that operator, no gradient flows to it -- the extent is not differentiable)."""
    if covars is not None:
        raise NotImplementedError("precomputed covariances are not supported; pass quats+scales")
    if packed:
        raise NotImplementedError("packed=True is not supported (nerfstudio uses packed=False)")
    require_device(means, quats, scales, viewmats, Ks)
    means, quats, scales = _f32c(means), _f32c(quats), _f32c(scales)
    viewmats, Ks = _f32c(viewmats), _f32c(Ks)
    if means.dim() != 2 or means.shape[1] != 3 or quats.shape != (means.shape[0], 4) \
            or scales.shape != means.shape:
        raise ValueError("expected means [N,3], quats [N,4], scales [N,3]")
    if viewmats.dim() != 3 or viewmats.shape[1:] != (4, 4) or Ks.shape != (viewmats.shape[0], 3, 3):
        raise ValueError("expected viewmats [C,4,4], Ks [C,3,3]")
    rule = radius_rule_id(radius_rule)
    if opacities is not None:
        require_device(opacities)
        opacities = _f32c(opacities.detach())
        if opacities.shape != (means.shape[0],):
            raise ValueError("expected opacities [N]")
    return _Projection.apply(means, quats, scales, viewmats, Ks, int(width), int(height),
                             float(eps2d), float(near_plane), float(far_plane),
                             float(radius_clip), bool(calc_compensations), opacities if rule else None, rule)


class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, degree, dirs, coeffs, masks):
        n = dirs.shape[0]
        colors = torch.empty(n, 3, dtype=torch.float32, device=dirs.device)
        check(_lib.lib().mgs_sh_fwd(n, degree, coeffs.shape[1], ptr(dirs), ptr(coeffs), ptr(masks),
                                    ptr(colors), stream_handle()), "mgs_sh_fwd")
        ctx.save_for_backward(dirs, coeffs, masks)
        ctx.degree = degree
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        dirs, coeffs, masks = ctx.saved_tensors
        n = dirs.shape[0]
        v_coeffs = torch.empty_like(coeffs)
        v_dirs = torch.empty_like(dirs) if ctx.needs_input_grad[1] else None
        check(_lib.lib().mgs_sh_bwd(n, ctx.degree, coeffs.shape[1], ptr(dirs), ptr(coeffs),
                                    ptr(masks), ptr(_f32c(v_colors)), ptr(v_coeffs), ptr(v_dirs),
                                    stream_handle()), "mgs_sh_bwd")
        return None, v_dirs, v_coeffs, None


def spherical_harmonics(degrees_to_use: int, dirs: Tensor, coeffs: Tensor,
                        masks: Optional[Tensor] = None) -> Tensor:
    """colors[...,3] = sum_k Y_k(normalize(dirs)) coeffs[...,k,:] for k < (degrees_to_use+1)^2."""
    require_device(dirs, coeffs, masks)
    if not 0 <= degrees_to_use <= 3:
        raise ValueError(f"degrees_to_use={degrees_to_use} outside 0..3")
    if coeffs.shape[-2] < (degrees_to_use + 1) ** 2:
        raise ValueError("coeffs holds fewer than (degrees_to_use+1)^2 coefficients")
    if dirs.shape[:-1] != coeffs.shape[:-2] or dirs.shape[-1] != 3 or coeffs.shape[-1] != 3:
        raise ValueError("expected dirs [...,3] and coeffs [...,K,3] with equal batch dims")
    batch = dirs.shape[:-1]
    d = _f32c(dirs).reshape(-1, 3)
    c = _f32c(coeffs).reshape(-1, coeffs.shape[-2], 3)
    m = None
    if masks is not None:
        m = masks.reshape(-1).to(torch.uint8).contiguous()
    return _SphericalHarmonics.apply(int(degrees_to_use), d, c, m).reshape(*batch, 3)


@torch.no_grad()
def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, tile_size: int,
                tile_width: int, tile_height: int, sort: bool = True, packed: bool = False,
                n_cameras: Optional[int] = None, camera_ids: Optional[Tensor] = None,
                gaussian_ids: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """Tile intersection + sort.  Returns tiles_per_gauss [C,N] i32, isect_ids [n_isects] i64
    (cam | tile | depth bits, ascending), flatten_ids [n_isects] i32 (cam*N + gaussian).
    radii [C,N] (one radius: the square mean +- radius) or [C,N,2] (per-axis extents, gsplat >= 1.5).
    Reads the intersection count back to size the outputs (one sync, as the reference
    operator does); the render path in rendering.py avoids that with a capacity."""
    if packed or not sort:
        raise NotImplementedError("only packed=False, sort=True is supported")
    if tile_size != TILE_SIZE:
        raise NotImplementedError(f"tile_size must be {TILE_SIZE}")
    require_device(means2d, radii, depths)
    C, N = means2d.shape[0], means2d.shape[1]
    means2d, depths = _f32c(means2d), _f32c(depths)
    radii = radii.to(torch.int32)
    if radii.dim() == 3:
        if radii.shape[-1] != 2:
            raise ValueError("expected radii [C,N] or [C,N,2]")
        radii = radii.transpose(1, 2)                # planar [C,2,N] for the raw call
    radii = radii.contiguous()
    tpg, keys, ids = [], [], []
    for c in range(C):
        cap = max(1, int(_upper_bound_isects(radii[c], tile_width, tile_height)))
        tl = isect_tiles_raw(means2d[c], radii[c], depths[c], tile_width, tile_height, cap, c, C,
                             want_isect_ids=True)
        n = int(tl.n_isect.item())
        tpg.append(tl.tiles_per_gauss)
        keys.append(tl.isect_ids[:n])
        ids.append(tl.flatten_ids[:n] + c * N if c else tl.flatten_ids[:n])
    return torch.stack(tpg), torch.cat(keys), torch.cat(ids)


def _upper_bound_isects(radii_c: Tensor, tile_w: int, tile_h: int) -> int:
    """Cheap device-side bound: sum over visible Gaussians of min((2r/16+2)^2, tiles); radii_c [N] or planar [2,N]."""
    rx = radii_x(radii_c).clamp_min(0).to(torch.float32)
    ry = rx if radii_c.dim() == 1 else radii_c[1].clamp_min(0).to(torch.float32)
    side_x = torch.floor(2.0 * rx / TILE_SIZE) + 2.0
    side_y = torch.floor(2.0 * ry / TILE_SIZE) + 2.0
    per = torch.minimum(side_x.clamp_max(tile_w) * side_y.clamp_max(tile_h),
                        torch.tensor(float(tile_w * tile_h), device=rx.device))
    return int(torch.where(radii_x(radii_c) > 0, per, torch.zeros_like(per)).sum().item())


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int,
                        tile_height: int) -> Tensor:
    """First sorted index of every (camera, tile): int32 [C, tile_height, tile_width]."""
    require_device(isect_ids)
    isect_ids = isect_ids.contiguous()
    out = torch.empty(n_cameras, tile_height, tile_width, dtype=torch.int32,
                      device=isect_ids.device)
    check(_lib.lib().mgs_isect_offset_encode(isect_ids.numel(), ptr(isect_ids), n_cameras,
                                             tile_width, tile_height, ptr(out), stream_handle()),
          "mgs_isect_offset_encode")
    return out


class _RasterizeToPixels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, width, height,
                offsets_ext, flatten_ids, absgrad):
        C, N, ch = colors.shape
        tile_w, tile_h = -(-width // TILE_SIZE), -(-height // TILE_SIZE)
        n_tiles = tile_w * tile_h
        dev = means2d.device
        render = torch.empty(C, height, width, ch, dtype=torch.float32, device=dev)
        alphas = torch.empty(C, height, width, dtype=torch.float32, device=dev)
        last_ids = torch.empty(C, height, width, dtype=torch.int32, device=dev)
        for c in range(C):
            # ids in flatten_ids are cam*N + gaussian: hand the kernels the flat [C*N,...] views
            rasterize_fwd_raw(means2d.view(C * N, 2), conics.view(C * N, 3),
                              colors.view(C * N, ch), opacities.view(C * N),
                              backgrounds[c] if backgrounds is not None else None, width, height,
                              tile_w, tile_h, offsets_ext[c * n_tiles:], flatten_ids,
                              out=(render[c], alphas[c], last_ids[c]))
        ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, offsets_ext,
                              flatten_ids, alphas, last_ids)
        ctx.dims = (width, height, tile_w, tile_h, absgrad)
        return render, alphas.unsqueeze(-1)

    @staticmethod
    def backward(ctx, v_render, v_alphas):
        (means2d, conics, colors, opacities, backgrounds, offsets_ext, flatten_ids, alphas,
         last_ids) = ctx.saved_tensors
        width, height, tile_w, tile_h, absgrad = ctx.dims
        C, N, ch = colors.shape
        n_tiles = tile_w * tile_h
        dev = means2d.device
        v_render = _f32c(v_render)
        v_alphas = _f32c(v_alphas).reshape(C, height, width)
        acc = (torch.zeros(C * N, 2, device=dev), torch.zeros(C * N, 3, device=dev),
               torch.zeros(C * N, ch, device=dev), torch.zeros(C * N, device=dev),
               torch.zeros(C * N, 2, device=dev) if absgrad else None)
        for c in range(C):
            rasterize_bwd_raw(means2d.view(C * N, 2), conics.view(C * N, 3),
                              colors.view(C * N, ch), opacities.view(C * N),
                              backgrounds[c] if backgrounds is not None else None, width, height,
                              tile_w, tile_h, offsets_ext[c * n_tiles:], flatten_ids, alphas[c],
                              last_ids[c], v_render[c], v_alphas[c], absgrad, accum=acc)
        v_bg = None
        if backgrounds is not None and ctx.needs_input_grad[4]:
            v_bg = (v_render * (1.0 - alphas).unsqueeze(-1)).sum(dim=(1, 2))
        if absgrad:
            means2d.absgrad = acc[4].view(C, N, 2)
        return (acc[0].view(C, N, 2), acc[1].view(C, N, 3), acc[2].view(C, N, ch),
                acc[3].view(C, N), v_bg, None, None, None, None, None)


def rasterize_to_pixels(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor,
                        image_width: int, image_height: int, tile_size: int,
                        isect_offsets: Tensor, flatten_ids: Tensor,
                        backgrounds: Optional[Tensor] = None, masks: Optional[Tensor] = None,
                        packed: bool = False, absgrad: bool = False) -> Tuple[Tensor, Tensor]:
    """Depth-ordered alpha compositing.  means2d [C,N,2], conics [C,N,3], colors [C,N,ch],
    opacities [C,N], isect_offsets [C,th,tw], flatten_ids [n_isects] ->
    render_colors [C,H,W,ch], render_alphas [C,H,W,1]."""
    if packed:
        raise NotImplementedError("packed=True is not supported")
    if masks is not None:
        raise NotImplementedError("tile masks are not supported")
    if tile_size != TILE_SIZE:
        raise NotImplementedError(f"tile_size must be {TILE_SIZE}")
    require_device(means2d, conics, colors, opacities, isect_offsets, flatten_ids, backgrounds)
    C, N = means2d.shape[0], means2d.shape[1]
    ch = colors.shape[-1]
    if not 1 <= ch <= 32:
        raise ValueError(f"{ch} colour channels: supported range is 1..32")
    tile_w, tile_h = -(-image_width // TILE_SIZE), -(-image_height // TILE_SIZE)
    if tuple(isect_offsets.shape) != (C, tile_h, tile_w):
        raise ValueError(f"isect_offsets shape {tuple(isect_offsets.shape)} != {(C, tile_h, tile_w)}")
    flatten_ids = flatten_ids.to(torch.int32).contiguous()
    end = torch.full((1,), flatten_ids.numel(), dtype=torch.int32, device=means2d.device)
    offsets_ext = torch.cat([isect_offsets.reshape(-1).to(torch.int32), end])
    return _RasterizeToPixels.apply(_f32c(means2d), _f32c(conics), _f32c(colors),
                                    _f32c(opacities), _f32c(backgrounds), int(image_width),
                                    int(image_height), offsets_ext, flatten_ids, bool(absgrad))
