"""ctypes binding of libmgs.so (the C ABI declared in include/mgs.h).

There is no fallback: if the shared library is missing or fails to load, importing the ops
raises.  torch is imported first so that libmgs.so binds to the HIP runtime torch already
loaded (same soname, libamdhip64.so.7) and shares its streams and allocations.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_size_t, c_uint32, c_void_p, POINTER

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmgs.so")
# the same sources built with -DMGS_DEBUG_HOOKS: the only build that has the process-global mgs_debug_set_* knobs
DEBUG_LIB_PATH = os.path.join(_HERE, "csrc", "libmgs_debug.so")
DEBUG_HOOKS = ["mgs_debug_set_raster_cull", "mgs_debug_set_raster_opts", "mgs_debug_set_sort_opts"]

MGS_STATUS_ISECT_OVERFLOW = 1
MGS_VERSION = 430          # include/mgs.h this binding was written against (parameter lists change with it)


class MgsError(RuntimeError):
    pass


def _load(path: str = None, hooks: bool = False) -> ctypes.CDLL:
    LIB_PATH = path or globals()["LIB_PATH"]
    if not os.path.exists(LIB_PATH):
        raise MgsError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no "
            "CPU or PyTorch fallback for the render path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.mgs_version.restype = c_int
    have = lib.mgs_version()
    if have != MGS_VERSION:      # shifted parameter lists would end in a GPU fault, not in an error
        raise MgsError(f"{LIB_PATH} reports ABI version {have}, this binding was written for {MGS_VERSION} "
                       "(include/mgs.h): rebuild the library (`python robosimgs_amd/csrc/build.py --force`)")
    p, i, f, u32 = c_void_p, c_int, c_float, c_uint32
    sig = {
        "mgs_version": ([], c_int),
        "mgs_last_error_string": ([], c_char_p),
        "mgs_projection_fwd": ([i, p, p, p, p, p, i, i, f, f, f, f, p, p, p, p, p, p, i, p, p], c_int),
        "mgs_projection_bwd": ([i, p, p, p, p, p, i, i, f, p, p, p, p, p, p, p, p, p, p, p, p], c_int),
        "mgs_sh_fwd": ([i, i, i, p, p, p, p, p], c_int),
        "mgs_sh_bwd": ([i, i, i, p, p, p, p, p, p, p], c_int),
        "mgs_project_color_fwd": ([i, p, p, p, p, i, i, p, p, p, i, i, f, f, f, f, p, p, p, p, p, i, p, p, i, p, p, p, p], c_int),
        "mgs_project_color_bwd": ([i, p, p, p, p, i, i, p, p, p, i, i, f, p, p, i, i, p, p, p, p, p, p, p, p, p, p, p, p, i, p], c_int),
        "mgs_isect_tiles": ([i, p, p, p, p, p, p, i, i, i, i, i, u32, p, p, p, p, p, p, p, p, p, p, p, p, p, POINTER(c_size_t), p], c_int),
        "mgs_isect_offset_encode": ([u32, p, i, i, i, p, p], c_int),
        "mgs_render_frames": ([i, p, p, p, p, i, i, p, i, p, p, i, i, f, f, f, f, i, i, i, p, u32, p, p, p, p, p, p, i, p, p, POINTER(c_size_t), p], c_int),
        "mgs_train_state_layout": ([i, i, i, i, u32, i, i, POINTER(c_size_t), POINTER(c_size_t)], c_int),
        "mgs_render_frames_train": ([i, p, p, p, p, i, i, p, i, p, p, i, i, f, f, f, f, i, i, i, p, u32, i, p, p, p, p, POINTER(c_size_t), p], c_int),
        "mgs_render_frames_backward": ([i, p, p, p, p, i, i, p, i, p, p, i, i, f, i, i, i, p, u32, i, p, p, p, p, p, p, p, p, p, p, p, p, p, p, POINTER(c_size_t), p], c_int),
        "mgs_rasterize_fwd": ([i, p, p, p, p, p, p, i, i, i, i, i, p, p, p, i, p, p, p, p, i, p, p, i, p, p], c_int),
        "mgs_raster_checkpoint_floats": ([u32, i, i, i, i], c_size_t),
        "mgs_rasterize_bwd": ([i, p, p, p, p, p, i, i, i, i, i, p, p, p, p, p, p, p, p, p, p, p, p], c_int),
        "mgs_composite_over": ([i, p, p, p, p, p, p, p, p, p, p], c_int),
        "mgs_points_project": ([i, p, p, p, p, p, p], c_int),
        "mgs_points_depth_map": ([i, p, i, p, i, i, i, i, f, f, p, p, p, POINTER(c_size_t), p], c_int),
        "mgs_points_sample_mask": ([i, p, i, p, i, i, f, p, p, f, p, p], c_int),
        "mgs_frame_to_u8": ([i, p, i, p, p, p, p], c_int),
        "mgs_frame_to_dataset": ([i, i, p, i, p, p, p, p, p, i, p], c_int),
        "mgs_transform_gaussians": ([i, p, p, p, i, i, p, p, i, p, p, p, p, p, p, p], c_int),
        "mgs_l1_loss_fwd": ([c_size_t, p, p, p, p, POINTER(c_size_t), p], c_int),
        "mgs_l1_loss_bwd": ([c_size_t, p, p, p, p, p], c_int),
        "mgs_l1_loss_fwd_grad": ([c_size_t, p, p, p, p, p, POINTER(c_size_t), p], c_int),
        "mgs_l1_loss_bwd_scale": ([c_size_t, p, p, p], c_int),
        "mgs_rasterize_bwd_det": ([i, p, p, p, p, p, p, i, i, i, i, i, p, p, p, p, p, p, p, p, p, u32, p, p, i, i, p, p, p, p, p, p, POINTER(c_size_t), p], c_int),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(lib, name)          # AttributeError here == header/library mismatch
        fn.argtypes = argtypes
        fn.restype = restype
    if hooks:
        for name in DEBUG_HOOKS:
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = [c_int], None
        if os.environ.get("MGS_SORT_OPTS"):            # measurement knob: one-sweep radix passes (same lists)
            lib.mgs_debug_set_sort_opts(int(os.environ["MGS_SORT_OPTS"], 0))
        if os.environ.get("MGS_RASTER_OPTS"):          # measurement knob (scripts/, profiles/): never changes a pixel
            lib.mgs_debug_set_raster_opts(int(os.environ["MGS_RASTER_OPTS"], 0))
    return lib


_lib = None
_debug = None


def lib() -> ctypes.CDLL:
    """The library every op calls: libmgs.so, which has no process-global state.  MGS_USE_DEBUG_LIB=1 in the
    environment (measurement scripts that set MGS_SORT_OPTS / MGS_RASTER_OPTS) makes it libmgs_debug.so instead."""
    global _lib
    if _lib is None:
        _lib = debug_lib() if os.environ.get("MGS_USE_DEBUG_LIB") else _load()
    return _lib


def debug_lib() -> ctypes.CDLL:
    """libmgs_debug.so: the same sources with -DMGS_DEBUG_HOOKS, the only build that has mgs_debug_set_*."""
    global _debug
    if _debug is None:
        _debug = _load(DEBUG_LIB_PATH, hooks=True)
    return _debug


class use_debug_lib:
    """with use_debug_lib() as L: every op inside goes through libmgs_debug.so (tests of the hooks, A/B scripts)."""

    def __enter__(self):
        global _lib
        self._saved = _lib
        _lib = debug_lib()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


EXPORTS = ["mgs_version", "mgs_last_error_string", "mgs_projection_fwd", "mgs_projection_bwd",
           "mgs_sh_fwd", "mgs_sh_bwd", "mgs_project_color_fwd", "mgs_project_color_bwd",
           "mgs_isect_tiles", "mgs_isect_offset_encode", "mgs_rasterize_fwd", "mgs_rasterize_bwd",
           "mgs_rasterize_bwd_det", "mgs_composite_over", "mgs_points_project",
           "mgs_points_depth_map", "mgs_points_sample_mask", "mgs_l1_loss_fwd", "mgs_l1_loss_bwd", "mgs_l1_loss_fwd_grad", "mgs_l1_loss_bwd_scale",
           "mgs_transform_gaussians", "mgs_frame_to_u8", "mgs_frame_to_dataset", "mgs_render_frames",
           "mgs_raster_checkpoint_floats", "mgs_train_state_layout", "mgs_render_frames_train", "mgs_render_frames_backward"]


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().mgs_last_error_string().decode("utf-8", "replace")
        raise MgsError(f"{what} failed with status {rc}: {msg}")


def ptr(t) -> int | None:
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_handle() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MgsError("the render path runs on the GPU only: got a CPU tensor "
                           "(there is no CPU fallback; see oracle/ for the test-only checker)")
