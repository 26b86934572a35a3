"""Rendered frames in the on-disk layout the reference's tooling reads (the OUTPUT side of the path).

The reference ships the readers of a rendered dataset, not the writer:
  load_images  /root/reference/Articulation/utils/nerf2physic_utils.py:84-101  RGBA images, sorted by file name;
               `alpha > 0` is the object mask, masked-out pixels are repainted with `bg_change`;
  load_depths  :104-118  `[H,W,1]` arrays in `.npy.gz`, sorted by name, holding RAY DISTANCES that
               distance_to_depth (:135-146) turns back into z-depth (rays through integer pixel coordinates).
`frame_to_dataset` produces both on the device from an "RGB+ED" frame (one kernel, mgs_frame_to_dataset);
`DatasetWriter` puts them on disk: `<root>/images/frame_00000.png`, `<root>/depth/frame_00000.npy.gz`.
tests/golden/dataset_reference.npz holds what the reference's own readers return for files written here
(generator: tests/golden/make_dataset_golden.py).

The host half (PNG / npy.gz encoding, `unnormalize_points`) needs only numpy; the device half needs
torch and libmgs.so (no CPU fallback).
"""
from __future__ import annotations

import gzip
import io
import os
import struct
import zlib
from typing import Optional, Sequence, Tuple

import numpy as np


# ---- host half: files ---------------------------------------------------------------------------------
def _png_chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png_rgba(rgba: np.ndarray, level: int = 6) -> bytes:
    """8-bit RGBA PNG (colour type 6, filter 0 on every row): the byte stream is a function of the pixels and
    the zlib level only -- no timestamps, no ancillary chunks."""
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError(f"expected uint8 [H,W,4], got {a.shape}")
    h, w = a.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * 4)], axis=1).tobytes()
    return (b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
            + _png_chunk(b"IDAT", zlib.compress(raw, level)) + _png_chunk(b"IEND", b""))


def decode_png_rgba(data: bytes) -> np.ndarray:
    """Inverse of encode_png_rgba for the files this module writes (8-bit RGBA, filter 0 only)."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, w = 8, b"", 0
    h = 0
    while pos < len(data):
        (n,), tag = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            if (depth, ctype) != (8, 6):
                raise ValueError("only 8-bit RGBA")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 4 * w)
    if rows[:, 0].any():
        raise ValueError("only filter type 0")
    return rows[:, 1:].reshape(h, w, 4).copy()


def encode_npy_gz(a: np.ndarray, level: int = 9) -> bytes:
    """np.save of `a` inside a gzip stream without a time stamp (the bytes depend on the array and the level only;
    9 is gzip's default and what the golden files hold)."""
    raw = io.BytesIO()
    np.save(raw, np.ascontiguousarray(a))
    out = io.BytesIO()
    with gzip.GzipFile(fileobj=out, mode="wb", mtime=0, filename="", compresslevel=level) as f:
        f.write(raw.getvalue())
    return out.getvalue()


class DatasetWriter:
    """`<root>/images/<stem>_%05d.png` (RGBA) and `<root>/depth/<stem>_%05d.npy.gz` ([H,W,1] ray distance): the two
    directories the reference's load_images / load_depths list and sort.

    The files are zlib streams: at 1080p one frame is ~0.25 s of host time (level 6 PNG + level 9 gzip of the fp32 distance
    map), i.e. 4 frames/s on one thread behind a renderer that makes 4,300 (INTEGRATION.md has the measured table).
    workers > 0 encodes and writes on a thread pool (zlib releases the GIL): write() returns as soon as the frame is on
    the host, flush() / close() wait for the files; at most `max_pending` FILES (two per frame with a distance map) wait
    in memory.  png_level /
    gz_level trade file size for time; the defaults give the very bytes of tests/golden/."""

    def __init__(self, root: str, image_dir: str = "images", depth_dir: str = "depth", stem: str = "frame",
                 workers: int = 0, png_level: int = 6, gz_level: int = 9, max_pending: int = 64):
        self.image_dir, self.depth_dir, self.stem = os.path.join(root, image_dir), os.path.join(root, depth_dir), stem
        os.makedirs(self.image_dir, exist_ok=True)
        os.makedirs(self.depth_dir, exist_ok=True)
        self.png_level, self.gz_level = int(png_level), int(gz_level)
        self._pool, self._pending, self._max_pending = None, [], max(1, int(max_pending))
        if workers and workers > 0:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=int(workers))

    def _put(self, path: str, encode, array) -> None:
        def job():
            with open(path, "wb") as f:
                f.write(encode(array))
        if self._pool is None:
            job()
            return
        if isinstance(array, np.ndarray):                   # the caller may reuse its buffer before the worker gets to it
            array = np.array(array, copy=True)
        while len(self._pending) >= self._max_pending:      # bound the frames held in memory
            self._pending.pop(0).result()
        self._pending.append(self._pool.submit(job))

    def flush(self) -> None:
        """Wait until every frame handed to write() is on disk.  Every pending file is waited for, whatever fails; the
        first failure is re-raised afterwards."""
        first = None
        while self._pending:
            try:
                self._pending.pop(0).result()
            except BaseException as e:          # noqa: BLE001 -- kept and re-raised once nothing is pending
                first = first or e
        if first is not None:
            raise first

    def close(self) -> None:
        try:
            self.flush()
        finally:                                # a failed file must not leak the executor and its threads
            if self._pool is not None:
                self._pool.shutdown()
                self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def paths(self, index: int) -> Tuple[str, str]:
        name = f"{self.stem}_{int(index):05d}"
        return os.path.join(self.image_dir, name + ".png"), os.path.join(self.depth_dir, name + ".npy.gz")

    def write(self, index: int, rgba, distance=None) -> Tuple[str, Optional[str]]:
        """rgba: uint8 [H,W,4]; distance: [H,W] or [H,W,1] float32 / float64 (None: image only).  Tensors are
        copied to the host first."""
        img_path, dep_path = self.paths(index)
        a = _to_numpy(rgba)
        if a.ndim != 3 or a.shape[2] != 4:
            raise ValueError(f"expected uint8 [H,W,4], got {a.shape}")
        d = None
        if distance is not None:               # both arrays are validated before either file is queued
            d = _to_numpy(distance)
            if d.ndim == 2:
                d = d[:, :, None]
            if d.ndim != 3 or d.shape[2] != 1:
                raise ValueError(f"distance must be [H,W] or [H,W,1], got {d.shape}")
        self._put(img_path, lambda x: encode_png_rgba(x, self.png_level), a)
        if d is None:
            return img_path, None
        self._put(dep_path, lambda x: encode_npy_gz(x, self.gz_level), d)
        return img_path, dep_path


def read_dataset_frame(img_path: str, dep_path: Optional[str] = None):
    """What this module wrote: (rgba uint8 [H,W,4], distance [H,W] or None)."""
    with open(img_path, "rb") as f:
        rgba = decode_png_rgba(f.read())
    dist = None
    if dep_path is not None:
        with gzip.open(dep_path, "rb") as f:
            dist = np.load(f)[:, :, 0]
    return rgba, dist


def _to_numpy(x) -> np.ndarray:
    if isinstance(x, np.ndarray):
        return x
    return x.detach().cpu().numpy()          # a torch tensor: a fresh host copy (safe to encode later on a worker thread)


def unnormalize_points(points: np.ndarray, transform: np.ndarray, scale: float) -> np.ndarray:
    """Points of a nerfstudio export (e.g. `ns-export pointcloud`) back in the coordinates of transforms.json:
    p = T^-1 [scale^-1-homogeneous], the matrix algebra of the reference's load_ns_point_cloud
    (/root/reference/Articulation/utils/nerf2physic_utils.py:68-74) applied to a plain [N,3] array.
    transform: the 3x4 `transform` of dataparser_transforms.json, scale: its `scale`."""
    t = np.concatenate([np.asarray(transform, dtype=np.float64).reshape(3, 4), np.array([[0.0, 0.0, 0.0, 1.0 / scale]])], 0)
    inv = np.linalg.inv(t)
    p = np.asarray(points, dtype=np.float64)
    h = np.concatenate([p, np.ones((p.shape[0], 1))], axis=1) @ inv.T
    return h[:, :3] / h[:, 3:4]


# ---- device half --------------------------------------------------------------------------------------
def frame_to_dataset(colors, alphas, K=None, background: Optional[Sequence[float]] = None,
                     distance_dtype=None, want_rgba: bool = True, out=None):
    """colors [H,W,D] (first three channels RGB; for the distance map the LAST channel is the z-depth of an
    "RGB+ED" / "RGB+D" frame), alphas [H,W,1] or [H,W], K [3,3] (numpy / list; None = image only).
    Returns (rgba uint8 [H,W,4] or None, distance [H,W,1] or None) on the device.
    distance_dtype: torch.float32 (default, what `ns-render` stores; the reader recovers z to one ulp) or
    torch.float64 (the reader recovers z to the last fp32 bit), or torch.float16 (11 significant bits: the light
    payload of the multi-GPU gather).
    out = (rgba uint8 [H,W,4], distance [H,W,1]) writes into existing contiguous buffers (e.g. a gather's staging area)."""
    import torch

    from . import _lib
    from ._lib import check, ptr, require_device, stream_handle
    require_device(colors, alphas)
    if colors.dim() != 3:
        raise ValueError("colors must be [H,W,D]")
    h, w, d = colors.shape
    c = colors.to(torch.float32).contiguous()
    a = alphas.to(torch.float32).reshape(h, w).contiguous()
    dev = colors.device
    rgba = torch.empty(h, w, 4, dtype=torch.uint8, device=dev) if (want_rgba and out is None) else None
    dist, kinv = None, None
    if out is not None:
        rgba, dist_out = out
        if (rgba is not None and (rgba.dtype != torch.uint8 or not rgba.is_contiguous() or rgba.numel() != h * w * 4)) or \
                (dist_out is not None and (not dist_out.is_contiguous() or dist_out.numel() != h * w)):
            raise ValueError("out = (uint8 [H,W,4], float [H,W,1]), both contiguous")
        if dist_out is not None:
            distance_dtype = dist_out.dtype
    dist_type = {torch.float64: 1, torch.float16: 2}.get(distance_dtype, 0)
    if K is not None:
        if d < 4:
            raise ValueError("the distance map needs a depth channel: render with render_mode='RGB+ED'")
        if distance_dtype not in (None, torch.float32, torch.float64, torch.float16):
            raise ValueError("distance_dtype must be torch.float32, torch.float64 or torch.float16")
        kinv = np.ascontiguousarray(np.linalg.inv(np.asarray(K, dtype=np.float64).reshape(3, 3)))
        dist = (out[1] if out is not None and out[1] is not None
                else torch.empty(h, w, 1, dtype=distance_dtype or torch.float32, device=dev))
    bg = torch.tensor(list(background), dtype=torch.float32, device=dev) if background is not None else None
    check(_lib.lib().mgs_frame_to_dataset(w, h, ptr(c), d, ptr(a), ptr(bg),
                                          kinv.ctypes.data if kinv is not None else None, ptr(rgba), ptr(dist),
                                          dist_type, stream_handle()), "mgs_frame_to_dataset")
    return rgba, dist
