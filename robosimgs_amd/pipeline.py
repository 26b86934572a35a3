"""FrameRenderer: steady-state novel-view rendering of a static scene at HIP-graph speed.

A data-generation loop renders thousands of frames of ONE scene from changing cameras.  The
frame is a fixed sequence of six kernels whose sizes do not depend on the camera once the
tile-list capacity is fixed, so it is captured once per in-flight slot as a HIP graph whose
camera (viewmat, K) lives in device buffers that are overwritten before each replay.  Several
slots, each on its own stream, keep independent frames in flight: the latency-bound binning
kernels of one frame run under the VALU-bound raster of another (+37 % frames/s at 1 M
Gaussians, 1080p on MI355X with 3 slots).

    r = FrameRenderer(gaussians.to_torch("cuda"), 1920, 1080, sizing_camera=(vm0, K0))
    tickets = [r.submit(cam.viewmat(), cam.K) for cam in cams[:3]]
    for cam in cams[3:]:
        t = tickets.pop(0)
        frame = r.fetch(t)                       # dict(colors, alphas): the slot's own buffers
        consume(frame)                           # enqueue the reads on the current stream ...
        r.release(t)                             # ... then hand the slot back
        tickets.append(r.submit(cam.viewmat(), cam.K))
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .rendering import rasterization

# ---- streams for the in-flight slots: one hardware queue each, none of them the consumer's -------------------------
# HIP multiplexes a process's streams onto a few hardware queues (four by default, GPU_MAX_HW_QUEUES).  Two slots whose
# streams share a queue cannot overlap their frames, and a slot that shares the queue of the CONSUMER stream (the one
# fetch() makes wait for a frame) cannot start its next frame until the frame being fetched is done: with three slots
# either collision costs 14 % (3,950 instead of 4,580 frames/s at 1 M Gaussians -- the rate of two frames in flight).
# Which torch stream lands on which queue depends on how many streams the process made before: the first two renderers
# of a process happened to get clean sets, the third did not (profiles/r4/00_experiments.md section 15).  So the sets
# are not left to chance: candidate streams are classed by hardware queue with pairs of spin kernels (two kernels on
# one queue take twice as long as on two) and the slots get streams of distinct queues other than the consumer's.
# One probe per device and consumer stream (~10 ms), cached for the life of the process.
_SLOT_STREAMS: Dict = {}
_SPIN_CYCLES = 600_000          # ~0.25 ms at the shader clock


def _spin_pair_ms(a, b, dev) -> float:
    """Host-timed: one spin kernel on stream a, one on stream b, until both are done."""
    import time
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s in (a, b):
        with torch.cuda.stream(s):
            torch.cuda._sleep(_SPIN_CYCLES)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3


def independent_streams(dev, n: int, candidates: int = 24) -> List:
    """`n` streams for the in-flight slots of a renderer on device `dev`: on pairwise different hardware queues, none on
    the queue of the current (consumer) stream, as far as the process's hardware queues allow (then the remaining slots
    share among themselves, never with the consumer).  Probed once per (device, consumer stream), reused by every
    renderer of the process."""
    dev = torch.device(dev)
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream)
    have = _SLOT_STREAMS.setdefault(key, {"reps": [], "pool": [], "single_ms": None})
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("FrameRenderer cannot be built inside a graph capture")
    if have["single_ms"] is None:
        with torch.cuda.stream(cur):
            torch.cuda._sleep(1000)                       # (first launch of the spin kernel: module load)
        have["single_ms"] = min(_spin_pair_ms(cur, cur, dev) for _ in range(2)) / 2.0

    def shares_queue(a, b) -> bool:
        return min(_spin_pair_ms(a, b, dev) for _ in range(2)) > 1.6 * have["single_ms"]

    while len(have["reps"]) < n and len(have["pool"]) < candidates:
        s = torch.cuda.Stream(dev)
        with torch.cuda.stream(s):
            torch.cuda._sleep(1000)                       # first use binds the stream to its hardware queue
        have["pool"].append(s)
        if shares_queue(cur, s) or any(shares_queue(r, s) for r in have["reps"]):
            continue
        have["reps"].append(s)
    reps = have["reps"]
    if not reps:                                          # one hardware queue for everything: nothing to choose
        reps = have["pool"][:1] or [torch.cuda.Stream(dev)]
    return [reps[i % len(reps)] for i in range(n)]


def locality_order(means: torch.Tensor, bits: int = 10) -> torch.Tensor:
    """Permutation (int64 [N]: new position -> index in `means`) that puts the Gaussians in Morton (Z-curve) order of
    their means: neighbours in space become neighbours in memory.  Same ordering as Gaussians.sorted_by_locality()."""
    m = means.detach().to(torch.float32)
    lo, hi = m.min(dim=0).values, m.max(dim=0).values
    q = ((m - lo) / (hi - lo).clamp_min(1e-30) * float((1 << bits) - 1)).to(torch.int64).clamp_(0, (1 << bits) - 1)
    code = torch.zeros(m.shape[0], dtype=torch.int64, device=m.device)
    for b in range(bits):
        for axis in range(3):
            code |= ((q[:, axis] >> b) & 1) << (3 * b + axis)
    return torch.argsort(code, stable=True)


class FrameRenderer:
    def __init__(self, tensors: Dict, width: int, height: int, render_mode: str = "RGB",
                 frames_in_flight: int = 3, isect_capacity: Optional[int] = None,
                 capacity_margin: float = 1.5, background: Optional[torch.Tensor] = None,
                 sizing_camera=None, group_ids: Optional[torch.Tensor] = None, n_groups: int = 0,
                 rotate_sh: bool = True, reorder: Optional[str] = "morton", dataset_output=None, dataset_K=None,
                 dataset_keep_float: bool = False, **raster_kw):
        """tensors: dict(means, quats, scales, opacities, colors, sh_degree) on the GPU
        (Gaussians.to_torch()); `self.t` is the renderer's own (by default Morton-ordered) copy.  isect_capacity: slots reserved for tile intersections per
        frame; if None it is measured once with `sizing_camera` = (viewmat, K) (required then)
        and multiplied by `capacity_margin`.  A frame that needs more raises on fetch().

        dataset_output (torch.float16 / float32 / float64) with dataset_K (the 3x3 intrinsics every camera of the run
        shares; render_mode "RGB+ED"): the frames leave the RASTER as the dataset frames the reference reads -- RGBA8 +
        ray distance, byte for byte what dataset.frame_to_dataset makes of the float frame -- in the slot's own buffer;
        fetch() returns "rgba" [H,W,4] u8, "distance" [H,W,1] and "dataset" (both as one flat byte buffer: ONE copy puts a
        frame into a gather's staging area).  6 - 12 bytes per pixel leave the kernel instead of 20 and no conversion
        pass reads them back.  dataset_keep_float=False (default): the float frame is not written at all and fetch()
        returns colors = alphas = None; True writes both.

        Dynamic scenes (articulated parts, a moving robot): give group_ids (int32 [N] on the GPU,
        -1 = static) and n_groups; submit(..., rotations=, translations=[, scales=]) then poses
        the groups for that frame.  Every slot owns a posed copy of the Gaussians and its graph
        starts with mgs_transform_gaussians(rest pose -> copy) reading the slot's transform buffer,
        so a posed frame costs one small upload and 14-97 us of GPU time more than a static one."""
        # reorder="morton" (default): the renderer keeps ITS OWN copy of the scene in Morton order of the means -- a
        # one-off at construction, like any acceleration structure of a static scene.  Frames carry nothing per
        # Gaussian, so nothing has to be mapped back; the image is the one the caller's order gives except where two
        # Gaussians of a pixel tie in depth to the last bit (ties go by index).  With three frames in flight the sorted
        # scene renders 4,400 instead of 3,880 frames/s at 1 M Gaussians: a tile's list entries gather from a narrow
        # index range, a binning workgroup's pairs fall into few tile groups, culled Gaussians are culled by the wave.
        # `self.order` maps the renderer's index to the caller's; reorder=None keeps the caller's order.
        if reorder not in (None, "morton"):
            raise ValueError(f"reorder {reorder!r} not in (None, 'morton')")
        self.order = None
        if reorder == "morton" and tensors["means"].shape[0] > 1:
            self.order = locality_order(tensors["means"])
            n = tensors["means"].shape[0]
            tensors = {k: (v.index_select(0, self.order).contiguous()
                           if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n else v) for k, v in tensors.items()}
            if group_ids is not None:
                group_ids = group_ids.index_select(0, self.order)
        self.t = tensors
        self.group_ids = group_ids.to(torch.int32).contiguous() if group_ids is not None else None
        self.n_groups = int(n_groups)
        self.rotate_sh = bool(rotate_sh) and int(tensors.get("sh_degree") or 0) >= 1
        if self.group_ids is not None and self.n_groups < 1:
            raise ValueError("group_ids needs n_groups >= 1")
        self.dev = tensors["means"].device
        self.width, self.height, self.mode = int(width), int(height), render_mode
        self.kw = dict(raster_kw)
        self.dataset_dtype, self.dataset_K, self.dataset_keep_float = dataset_output, None, bool(dataset_keep_float)
        if dataset_output is not None:
            if dataset_K is None or render_mode != "RGB+ED":
                raise ValueError("dataset_output needs dataset_K (the shared 3x3 intrinsics) and render_mode='RGB+ED'")
            if dataset_output not in (torch.float16, torch.float32, torch.float64):
                raise ValueError("dataset_output must be torch.float16, torch.float32 or torch.float64")
            self.dataset_K = np.asarray(dataset_K, dtype=np.float64).reshape(3, 3)
        # several frames in flight: total work matters, not one launch's duration (rendering.py) -- but a frame that is
        # submitted while NO other frame is in flight (a synchronous render(), the first frame of a burst) has the GPU to
        # itself and takes the other schedule: every slot holds both graphs (one memory pool), submit() picks.  The two
        # schedules give the same pixels bit for bit.  A caller who names a raster_schedule gets that one only.
        self._both = "raster_schedule" not in self.kw and int(frames_in_flight) > 1
        self.kw.setdefault("raster_schedule", "throughput" if int(frames_in_flight) > 1 else "latency")
        # the slots' frames keep no per-Gaussian arrays nobody reads (rendering.py: lean_meta)
        self.kw.setdefault("lean_meta", True)
        self.bg = background
        if isect_capacity is None:
            if sizing_camera is None:
                raise ValueError("give isect_capacity or a sizing_camera=(viewmat, K)")
            vm, K = self._cam_tensors(*sizing_camera)
            _, _, meta = self._raster(vm, K, None)
            isect_capacity = int(int(meta["n_isects"].max().item()) * capacity_margin) + 4096
        self.capacity = int(isect_capacity)
        self.n_slots = max(1, int(frames_in_flight))
        self._slots: List[Dict] = []
        streams = independent_streams(self.dev, self.n_slots)     # one hardware queue per slot, none the consumer's
        for i in range(self.n_slots):
            self._slots.append(self._capture_slot(streams[i]))
        self._next = 0

    # -- internals ---------------------------------------------------------------------
    def _cam_tensors(self, viewmat, K):
        vm = torch.as_tensor(np.asarray(viewmat, dtype=np.float32)).reshape(1, 4, 4).to(self.dev)
        Kt = torch.as_tensor(np.asarray(K, dtype=np.float32)).reshape(1, 3, 3).to(self.dev)
        return vm, Kt

    def _raster(self, vm, K, cap, t=None, dataset_out=None, schedule=None):
        t = self.t if t is None else t
        kw = self.kw if schedule is None else dict(self.kw, raster_schedule=schedule)
        return rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], vm,
                             K, self.width, self.height, sh_degree=t.get("sh_degree"),
                             render_mode=self.mode, backgrounds=self.bg, isect_capacity=cap,
                             dataset_out=dataset_out, **kw)

    def _capture_slot(self, stream) -> Dict:
        # the slot's camera: one 25-float device buffer (viewmat | K), so a submit is ONE small copy
        cam = torch.zeros(32, device=self.dev)
        vm, K = cam[:16].view(1, 4, 4), cam[16:25].view(1, 3, 3)
        vm.copy_(torch.eye(4, device=self.dev).reshape(1, 4, 4))
        vm[0, 2, 3] = -1e3                                 # warm-up camera: everything is behind it, nothing to bin
        K.copy_(torch.tensor([[[1.0, 0, 0.5], [0, 1.0, 0.5], [0, 0, 1]]], device=self.dev))
        pose = None
        if self.group_ids is not None:       # dynamic scene: this slot's transforms and posed copy
            from .transform import pack_transforms
            eye = np.tile(np.eye(3), (self.n_groups, 1, 1))
            x0, r0 = pack_transforms(eye, np.zeros((self.n_groups, 3)), None, 3 if self.rotate_sh else 0)
            pose = {"x": torch.from_numpy(x0).to(self.dev),
                    "r": torch.from_numpy(r0).to(self.dev) if r0 is not None else None,
                    "t": {k: (v.clone() if torch.is_tensor(v) and k in ("means", "quats", "scales", "colors") else v)
                          for k, v in self.t.items()}}

        ds, ds_out = None, None
        if self.dataset_dtype is not None:          # the slot's dataset frame: [RGBA8 plane | distance plane] as bytes
            n_px = self.width * self.height
            esz = torch.empty(0, dtype=self.dataset_dtype).element_size()
            # the distance plane starts n_px * 4 bytes into the frame: with float64 distances and an odd pixel count that
            # is 4 mod 8, so the frame starts 4 bytes into its allocation there (the RGBA plane has no alignment to keep)
            lead = (-(n_px * 4)) % esz
            flat = torch.empty(lead + n_px * (4 + esz), dtype=torch.uint8, device=self.dev)[lead:]
            ds = {"dataset": flat, "rgba": flat[:n_px * 4].view(self.height, self.width, 4),
                  "distance": flat[n_px * 4:].view(self.dataset_dtype).view(self.height, self.width, 1)}
            ds_out = (ds["rgba"].unsqueeze(0), ds["distance"].unsqueeze(0), self.dataset_K, self.dataset_keep_float)

        def body(schedule):
            if pose is None:
                return self._raster(vm, K, self.capacity, dataset_out=ds_out, schedule=schedule)
            from .transform import transform_gaussians
            posed = transform_gaussians(self.t, group_ids=self.group_ids, rotate_sh=self.rotate_sh,
                                        out=pose["t"], packed=(pose["x"], pose["r"]))
            return self._raster(vm, K, self.capacity, posed, dataset_out=ds_out, schedule=schedule)
        variants, pool = {}, None
        for schedule in (("throughput", "latency") if self._both else (self.kw["raster_schedule"],)):
            with torch.cuda.stream(stream):
                for _ in range(2):
                    body(schedule)
                torch.cuda.synchronize(self.dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream, pool=pool):       # (a slot's graphs never run together: one pool)
                    colors, alphas, meta = body(schedule)
                pool = graph.pool()
            torch.cuda.synchronize(self.dev)
            # "replayed": a captured graph has recorded its kernels, not run them -- until its first replay its outputs
            # (the overflow word among them) are whatever the graph pool's memory held
            variants[schedule] = {"graph": graph, "colors": colors, "alphas": alphas, "meta": meta, "replayed": False}
        first = variants[self.kw["raster_schedule"]]
        return {"stream": stream, "vm": vm, "K": K, "cam": cam, "pose": pose, "variants": variants, "variant": self.kw["raster_schedule"],
                "graph": first["graph"], "colors": first["colors"], "alphas": first["alphas"], "meta": first["meta"],
                "done": torch.cuda.Event(), "released": torch.cuda.Event(), "state": "free", "ds": ds}

    # -- API ---------------------------------------------------------------------------
    @staticmethod
    def pack_camera(viewmat, K, device=None) -> torch.Tensor:
        """(viewmat | K) as one 25-float tensor; submit(packed, None) then uploads it with a single copy."""
        v = torch.as_tensor(np.asarray(viewmat, dtype=np.float32) if not torch.is_tensor(viewmat) else viewmat)
        k = torch.as_tensor(np.asarray(K, dtype=np.float32) if not torch.is_tensor(K) else K)
        out = torch.cat([v.reshape(16).float(), k.reshape(9).float()])
        return out.to(device) if device is not None else out

    def submit(self, viewmat, K=None, rotations=None, translations=None, scales=None) -> int:
        """Enqueue one frame (viewmat: OpenCV world-to-camera 4x4, K: 3x3; numpy or tensors; or
        viewmat = pack_camera(viewmat, K) and K = None).  Dynamic scenes: rotations [G,3,3],
        translations [G,3] (and uniform scales [G]) pose the groups for this frame; omitted, the
        slot keeps the pose of its previous frame (the rest pose at first).
        Returns a ticket for fetch().  Slots are used round-robin: the slot's previous frame
        must have been fetched and released."""
        slot = self._next
        s = self._slots[slot]
        if s["state"] != "free":
            raise RuntimeError(f"slot {slot} still holds a frame that was not released "
                               f"({self.n_slots} frames in flight at most)")
        self._next = (slot + 1) % self.n_slots
        # nothing else in flight: this frame has the GPU to itself -- the schedule with the shortest launch
        want = "latency" if (self._both and all(o["state"] == "free" for o in self._slots)) else self.kw["raster_schedule"]
        if s["variant"] != want:
            v = s["variants"][want]
            s.update(variant=want, graph=v["graph"], colors=v["colors"], alphas=v["alphas"], meta=v["meta"])
        packed = None
        if self.dataset_K is not None:
            # the dataset epilogue turns depth into ray distance with the intrinsics the renderer was built with (they
            # are arguments of the captured kernel): a frame submitted with another K would carry wrong distances.
            # Checked wherever the submitted K is on the host; a K on the device is the caller's to keep equal.
            k_host = None
            if K is not None and not (torch.is_tensor(K) and K.is_cuda):
                k_host = np.asarray(K, dtype=np.float64).reshape(3, 3)
            elif K is None and torch.is_tensor(viewmat) and not viewmat.is_cuda:
                k_host = viewmat.reshape(-1)[16:25].double().numpy().reshape(3, 3)
            if k_host is not None and not np.allclose(k_host, self.dataset_K, rtol=1e-6, atol=0.0):
                raise ValueError("this FrameRenderer writes dataset frames for the intrinsics dataset_K it was built with; "
                                 "the submitted K differs (build one renderer per set of intrinsics)")
        if torch.is_tensor(viewmat) and K is None:
            packed = viewmat.reshape(-1)               # pack_camera(): (viewmat | K) already on one tensor
        elif not torch.is_tensor(viewmat) and not torch.is_tensor(K):
            packed = torch.from_numpy(np.concatenate([np.asarray(viewmat, dtype=np.float32).reshape(16),
                                                      np.asarray(K, dtype=np.float32).reshape(9)]))
        with torch.cuda.stream(s["stream"]):
            s["stream"].wait_event(s["released"])      # the previous consumer's reads are done
            if packed is not None:
                s["cam"][:25].copy_(packed, non_blocking=True)
            else:
                s["vm"].copy_(torch.as_tensor(viewmat).reshape(1, 4, 4), non_blocking=True)
                s["K"].copy_(torch.as_tensor(K).reshape(1, 3, 3), non_blocking=True)
            if rotations is not None:
                if s["pose"] is None:
                    raise ValueError("this FrameRenderer was built without group_ids: the scene is static")
                from .transform import pack_transforms
                x, r = pack_transforms(rotations, translations, scales, 3 if self.rotate_sh else 0)
                if x.shape[0] != self.n_groups:
                    raise ValueError(f"{x.shape[0]} transforms for {self.n_groups} groups")
                s["pose"]["x"].copy_(torch.from_numpy(x), non_blocking=True)
                if r is not None:
                    s["pose"]["r"].copy_(torch.from_numpy(r), non_blocking=True)
            s["graph"].replay()
            s["variants"][s["variant"]]["replayed"] = True
            s["done"].record(s["stream"])
        s["state"] = "submitted"
        return slot

    def fetch(self, ticket: int, check: bool = True) -> Dict:
        """Make the frame of `ticket` visible to the current stream and return
        dict(colors [H,W,D], alphas [H,W,1], meta).  The tensors are the slot's own buffers:
        enqueue whatever reads them on the current stream, then call release(ticket).
        check=True reads the slot's overflow word back (one 4-byte sync) and raises if the tile
        lists did not fit."""
        s = self._slots[ticket]
        if s["state"] != "submitted":
            raise RuntimeError(f"ticket {ticket} has no frame in flight")
        torch.cuda.current_stream(self.dev).wait_event(s["done"])
        s["state"] = "fetched"
        if check and bool((s["meta"]["isect_status"] != 0).any().item()):
            need = int(s["meta"]["n_isects"].max().item())
            s["released"].record(torch.cuda.current_stream(self.dev))   # the slot is usable again
            s["state"] = "free"
            raise _lib.MgsError(f"frame needs {need} tile intersections, capacity is "
                                f"{self.capacity}: build the FrameRenderer with a larger "
                                "isect_capacity / capacity_margin")
        out = {"colors": s["colors"][0], "alphas": s["alphas"][0], "meta": s["meta"]}
        if s["ds"] is not None:
            out.update(s["ds"])
            if not self.dataset_keep_float:          # the float frame was never written
                out["colors"] = out["alphas"] = None
        return out

    def isect_status_max(self) -> int:
        """Largest overflow status word over every slot and both of its graphs (reads them back: call it outside timed
        regions).  0 = no frame rendered so far needed more tile intersections than the capacity.  Only graphs that have
        been replayed are read: a variant that never ran has never written its status word."""
        return max((int(v["meta"]["isect_status"].max().item()) for s in self._slots for v in s["variants"].values()
                    if v["replayed"]), default=0)

    def release(self, ticket: int) -> None:
        """Hand the slot back: its next frame will start after everything enqueued so far on the
        current stream (the consumer of the fetched tensors)."""
        s = self._slots[ticket]
        if s["state"] != "fetched":
            raise RuntimeError(f"ticket {ticket} was not fetched")
        s["released"].record(torch.cuda.current_stream(self.dev))
        s["state"] = "free"

    def to_caller_order(self, x: torch.Tensor, dim: int = 0) -> torch.Tensor:
        """A per-Gaussian array of a frame's meta (radii, means2d, depths ... with lean_meta=False; the renderer's own
        index order along `dim`) in the order of the tensors the renderer was built from.  Identity when reorder=None."""
        if self.order is None:
            return x
        out = torch.empty_like(x)
        out.index_copy_(dim, self.order.to(x.device), x)
        return out

    def render(self, viewmat, K) -> Dict:
        """Synchronous convenience: one frame, returned as copies (the slot is released).  A renderer built with
        dataset_output= returns "rgba" / "distance" (and colors / alphas only with dataset_keep_float=True, else None)."""
        t = self.submit(viewmat, K)
        f = self.fetch(t)
        out = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in f.items() if k not in ("meta", "dataset")}
        out["meta"] = f["meta"]
        self.release(t)
        return out

    def render_sequence(self, cameras, consume) -> None:
        """Render Camera objects (robosimgs_amd.Camera) keeping all slots busy;
        `consume(index, frame)` is called in order on the current stream."""
        cams = list(cameras)
        tickets: List = []
        nxt = 0
        for i in range(len(cams)):
            while nxt < len(cams) and len(tickets) < self.n_slots:
                tickets.append(self.submit(cams[nxt].viewmat(), cams[nxt].K))
                nxt += 1
            t = tickets.pop(0)
            consume(i, self.fetch(t))
            self.release(t)
