#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1 (BASELINE.json configs[1]): a step is one forward render of the 1 M-Gaussian, SH-degree-3
synthetic scene at 1920x1080 in the mode the north star describes -- RGB + expected depth + alpha
(`render_mode="RGB+ED"`): projection+SH, tile binning, tile raster, all through the C ABI of
libmgs.so, replayed as one HIP graph with the scene resident in HBM.

N > 1 (configs[3] as BASELINE.json writes it; one process per GPU over RCCL -- started by torch.distributed.run as the
driver does, or by bench.py itself when it is called as plain `python bench.py --gpus N` without WORLD_SIZE in the
environment): a step is one pass over the 64-camera novel-view ring theta_k = 2 pi k / 64, EQUAL contiguous blocks per rank
(8 views per GPU at 8 ranks: rank r renders k in [8r, 8r + 8)), and rank 0 gathers all 64 finished frames.
  * `value` is measured with the payload that carries the north star's 1e-4 to the root: the raw fp32 renders, RGB +
    expected depth + alpha, 20 bytes per pixel = 41.5 MB per frame (`--gather-dtype fp32`, the default).
  * Rank 0 receives (N - 1) / N of every frame over its seven xGMI links, which bounds the whole job at (its inbound
    rate) / (payload per frame) whatever the renderers do: with fp32 frames a single root cannot reach 6 x the one-GPU
    rate (config.root_bound).  The SAME run therefore times the ring again in three named alternates
    (config.alternates; `--one-payload` skips them), each a smaller thing to move:
      dataset                  RGBA8 + fp32 ray distance, 8 B per pixel: the layout DatasetWriter stores and the
                               reference's load_images / load_depths read (reference-pinned, tests/golden/); equal blocks
      dataset16                RGBA8 + fp16 ray distance, 6 B per pixel (QUANTISED: 11 significant bits of distance,
                               ~3 mm at 7 units; outside 1e-4); equal blocks
      dataset16_weighted_root  the same payload with rank 0 rendering a smaller block the more ranks send to it
                               (`--alt-root-weight`, default max(0.5, 1 - 0.07 (N - 1)): 31 + 33 cameras at 2 ranks,
                               13 + 17 + 17 + 17 at 4, 4 + 9 + 9 + 9 + 9 + 8 + 8 + 8 at 8) -- the tuned variant, NOT configs[3]'s
                               "8 views per GPU"
    Dataset payloads leave the raster itself (FrameRenderer(dataset_output=)); nothing is converted afterwards.
  * config.per_rank carries, per leg, each rank's HIP-event split of its last timed region (render span, conversions,
    what its stream still waited for behind its last conversion), so that a scaling run shows where the time went.
Total work is fixed as N grows: scaling is "strong"; value = frames all ranks rendered / time.

Timing: W warm-up steps, then regions of EXACTLY K steps, each bracketed by barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over ranks; regions are repeated until
0.5 s have been timed (a single 20-frame region is 7 ms: too short to be stable) and the MEDIAN
region gives ms_per_step and value.

Rank 0 prints ONE JSON line.  Beside the contract fields it carries
  roofline      tile-raster forward kernel: algorithmic bytes / HIP-event time vs 8 TB/s (+ roofline_projection, _binning)
  cpu_baseline  the C++/OpenMP port in oracle/gs_cpu.cpp timed on this box's host cores
  fwd_bwd       the training-step variant (configs[2]): forward + L1 loss + backward, in the caller's and in Morton order,
                with fwd_bwd.roofline for its dominant kernel (the segmented backward raster)
  stress_4k     configs[4]: 5 M Gaussians at 3840x2160, per stage against its algorithmic bytes, and frames/s
  heavy_tailed  NOT a BASELINE config: the frame and the training step on a clustered, heavy-tailed 1 M-Gaussian scene
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from robosimgs_amd import camera_ring, synthetic_scene  # noqa: E402
from robosimgs_amd import ops  # noqa: E402
from robosimgs_amd.distributed import root_weights, shard_cameras, shard_sizes  # noqa: E402
from robosimgs_amd.rendering import rasterization  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)
MODE = "RGB+ED"                # what splatfacto renders and INTEGRATION.md tells users to call
RING = 64                      # configs[3]: 64 novel-view cameras
PAYLOADS = {"dataset16": "dataset frames, light and QUANTISED: RGBA8 + fp16 ray distance, 6 B per pixel -- the RGBA image load_images "
                         "reads and a half-precision distance map (np.float16: load_depths / distance_to_depth take it as it is; 11 "
                         "significant bits, ~3 mm at 7 units: outside the north star's 1e-4); written by the raster's epilogue",
            "dataset": "dataset frames: RGBA8 + fp32 ray distance, 8 B per pixel, the layout DatasetWriter stores and the "
                       "reference's load_images / load_depths read (reference-pinned); written by the raster's epilogue",
            "fp32": "fp32 RGB + expected depth + alpha (20 B per pixel): the raw renders, the only payload that carries 1e-4 to the root",
            "u8": "8-bit RGB images (frame_to_u8 on the device, inside the timed region)"}
PAYLOAD_BYTES_PER_PX = {"dataset16": 6, "dataset": 8, "fp32": 20, "u8": 3}
XGMI_LINK_GBS = 50.0           # ASSUMED: what an RCCL point-to-point gather sustains per xGMI link and direction (the guide's
                               # peak is 153 GB/s per link both ways = 76.5 per direction; nothing here can measure it: no multi-GPU
                               # box).  It only feeds the `root_bound` table of the JSON line, no default and no `value`.


def root_bound_frames_per_s(mode, W, H, world):
    """Ceiling the single gathering rank puts on the job: its inbound links carry (world - 1) / world of every frame."""
    if world < 2:
        return None
    inbound = XGMI_LINK_GBS * 1e9 * min(world - 1, 7)
    return inbound / (PAYLOAD_BYTES_PER_PX[mode] * W * H * (world - 1) / world)


def scaling_prediction(rate_by_payload, W, H, worlds=(2, 4, 8)):
    """What the first SCALE record can be read against: per payload and world size N, the ring's predicted frames/s
        min(N x this rank's measured rate with that payload, the single root's ceiling root_bound_frames_per_s)
    rate_by_payload: {payload: frames/s of ONE rank rendering AND delivering that payload, measured in this run} (a world of
    one: the float frame for "fp32", the dataset frames written by the raster for the dataset payloads where that leg ran,
    otherwise the fp32 rate as a stand-in, flagged).  Not in the model: the root also renders while RCCL's receive kernels
    write (N - 1) / N of every frame into its HBM (a world of one with --force-gather reads 4-5 % for dataset frames, 9-10 % for
    fp32 frames), and the assumed link rate XGMI_LINK_GBS.  speedup = predicted / the fp32 one-rank rate."""
    base = rate_by_payload.get("fp32")
    out = {}
    for m, bpp in PAYLOAD_BYTES_PER_PX.items():
        r1 = rate_by_payload.get(m) or base
        if not r1:
            continue
        rows = {}
        for n in worlds:
            rb = root_bound_frames_per_s(m, W, H, n)
            pred = min(n * r1, rb)
            rows[str(n)] = {"render_bound": round(n * r1, 0), "root_bound": round(rb, 0), "predicted_frames_per_s": round(pred, 0),
                            "limited_by": "root" if rb < n * r1 else "render", "speedup_vs_one_rank_fp32": round(pred / base, 2) if base else None}
        out[m] = {"bytes_per_frame": bpp * W * H, "one_rank_frames_per_s": round(r1, 1),
                  "one_rank_rate_measured_with_this_payload": m in rate_by_payload, "by_world_size": rows}
    return out


PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--log-scale-mean", type=float, default=math.log(0.012))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--bwd-steps", type=int, default=30)
    ap.add_argument("--convert-on-consumer", action="store_true",
                    help="N > 1: render float frames and convert them to the dataset payload on the consumer stream between "
                         "fetch and release (the round-3 loop) instead of letting the raster write the payload "
                         "(FrameRenderer(dataset_output=)); world of one: 4,124-4,156 against 4,356-4,380 frames/s")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="repeat the K-step region until this much is timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stress", action="store_true", help="skip the configs[4] leg (5 M Gaussians at 3840x2160)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-reorder", action="store_true",
                    help="FrameRenderer keeps the scene in the order it is given (default: its own Morton-ordered copy)")
    # N > 1: what rank 0 collects in the leg `value` is measured on.  "fp32" (default) = the raw renders, RGB + expected depth
    # + alpha, 41.5 MB per frame (332 MB per rank and pass at 8 ranks) -- the payload that carries the north
    # star's 1e-4; "dataset" = RGBA8 + fp32 ray distance (8 B per pixel, what DatasetWriter stores); "dataset16" = RGBA8 + fp16
    # distance (6 B per pixel, quantised); "u8" = 8-bit RGB only.  The payloads not chosen here are timed as named alternates.
    ap.add_argument("--gather-dtype", choices=("dataset16", "dataset", "fp32", "u8"), default="fp32")
    # share of the ring the gathering rank renders, in units of the other ranks' (1 = equal blocks: configs[3]'s "8 views per
    # GPU", the default).  The alternate leg dataset16_weighted_root uses --alt-root-weight (default max(0.5, 1 - 0.07 (N - 1)) =
    # 0.93 / 0.79 / 0.51 at 2 / 4 / 8 ranks: 31 + 33; 13 + 17 + 17 + 17; 4 + 9 + 9 + 9 + 9 + 8 + 8 + 8 cameras).
    ap.add_argument("--root-weight", type=float, default=1.0)
    ap.add_argument("--alt-root-weight", type=float, default=None)
    ap.add_argument("--gather-batch", type=int, default=4, help="frames per collective (N > 1)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent frames in flight, one HIP stream + one HIP graph each")
    # debugging aid for the N > 1 control flow on a single-GPU box: all ranks share cuda:0 and the
    # collectives run over gloo with host staging.  Never used for a reported number.
    ap.add_argument("--debug-single-device-gloo", action="store_true")
    # debugging aid: run the N > 1 leg (camera ring + gather) in a world of one
    ap.add_argument("--force-gather", action="store_true")
    ap.add_argument("--one-payload", action="store_true", help="N > 1: time only the headline leg, not the named alternates")
    return ap.parse_args()


def self_launch(n, debug_gloo):
    """Re-run this command under torch.distributed.run with n ranks on this node; returns the exit code."""
    import socket
    import subprocess
    if not debug_gloo:
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, this node has {have} "
                  "(one rank per GPU; --debug-single-device-gloo shares cuda:0 for control-flow debugging only)",
                  file=sys.stderr)
            return 2
    with socket.socket() as sk:                          # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")               # (also keeps torchrun from printing its OMP notice)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)      # stderr passes through
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in r.stdout.splitlines():                      # anything else the ranks printed goes to stderr
        if l not in lines:
            print(l, file=sys.stderr)
    if r.returncode == 0 and len(lines) == 1:
        print(lines[0], flush=True)
        return 0
    print(f"bench.py: the {n}-rank run ended with exit code {r.returncode} and {len(lines)} result line(s)", file=sys.stderr)
    return r.returncode or 1


def barrier_sync(use_dist):
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()


def main():
    a = parse()
    if os.environ.get("BENCH_WATCHDOG_S"):
        # debugging aid for a stuck multi-rank run: every process dumps its Python stacks to stderr and exits after this
        # many seconds (the launcher then reports the ranks' exit code)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_WATCHDOG_S"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, what the driver's
        # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` does) and pass
        # rank 0's JSON line through as the only line on stdout
        raise SystemExit(self_launch(a.gpus, a.debug_single_device_gloo))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start bench.py with --gpus equal to the number of ranks")
    debug_gloo = a.debug_single_device_gloo
    if debug_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_gather
    out_fd = None
    if use_dist:
        # rank 0's JSON line must be the only thing on stdout: RCCL prints its version banner there (NCCL_DEBUG=VERSION
        # is set on the GPU boxes), so file descriptor 1 points at stderr from here on and the line goes to the saved one
        sys.stdout.flush()
        out_fd = os.dup(1)
        os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if debug_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ring = use_dist                       # the N > 1 workload: configs[3]'s camera ring

    W, H, deg = a.width, a.height, a.sh_degree
    scene = synthetic_scene(a.n, a.log_scale_mean, deg, seed=0)
    t = scene.to_torch(dev, deg)
    tile_w, tile_h = -(-W // 16), -(-H // 16)
    # cameras: the single theta = 0.3 view of configs[1], or this rank's block of the 64-camera ring.  Headline: equal blocks
    # (configs[3]: "sharded 8 views/GPU"); the alternate leg dataset16_weighted_root gives rank 0 a smaller block
    if a.alt_root_weight is None:
        a.alt_root_weight = max(0.5, 1.0 - 0.07 * (world - 1))
    sizing_cam = camera_ring(1, W, H, thetas=[0.3])[0]

    def cam_tensors(c):
        return (torch.from_numpy(c.viewmat().astype(np.float32)).to(dev)[None],
                torch.from_numpy(c.K.astype(np.float32)).to(dev)[None])

    def make_shard(root_weight):
        """This rank's cameras (and everybody's block sizes) for one weighting of the gathering rank."""
        from robosimgs_amd import FrameRenderer as FR_
        w_ = root_weights(world, root_weight) if (ring and world > 1) else None
        mine_ = shard_cameras(RING, world, rank, w_) if ring else range(1)
        thetas_ = [2.0 * math.pi * k / RING for k in mine_] if ring else [0.3]
        cams_ = camera_ring(len(thetas_), W, H, thetas=thetas_) if thetas_ else []
        sizes_ = shard_sizes(RING, world, w_) if ring else [1]
        return {"root_weight": root_weight if (ring and world > 1) else None, "weights": w_, "mine": mine_, "cams": cams_,
                "sizes": sizes_, "cam_devs": [FR_.pack_camera(*[x[0].contiguous() for x in cam_tensors(c)]) for c in cams_],
                # frames per collective: a per-frame collective costs ~80 us of launch / stream hand-over, a quarter of a frame
                "GB": max(1, min(a.gather_batch, max(1, max(sizes_)))) if ring else 1}

    shard = make_shard(a.root_weight)
    shard_alt = make_shard(a.alt_root_weight) if (ring and world > 1 and a.alt_root_weight != a.root_weight) else None
    cams = shard["cams"]

    vm, K = cam_tensors(sizing_cam)

    def forward(vm_, K_, cap=None, bounds="tight"):
        return rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                             vm_, K_, W, H, sh_degree=deg, render_mode=MODE, isect_capacity=cap,
                             tile_bounds=bounds)

    # sizing passes (read n_isect back, outside every timed region).  n_isect is the classic
    # mean +- radius count SURVEY.md 8(d) calibrates (5,019,7xx at config 2); the frames run on the
    # tightened lists (n_isect_binned, bit-identical image), so both are reported and priced.
    colors, alphas, meta = forward(vm, K, bounds="classic")
    torch.cuda.synchronize()
    n_isect = int(meta["n_isects"][0])
    n_vis = int((meta["radii"] > 0).sum())
    colors_t, alphas_t, meta_t = forward(vm, K)
    assert torch.equal(colors_t, colors) and torch.equal(alphas_t, alphas), "tight tile bounds changed the image"
    n_isect_binned = int(meta_t["n_isects"][0])
    del colors_t, alphas_t, meta_t, colors, alphas, meta
    need = n_isect_binned
    for c in cams + (shard_alt["cams"] if shard_alt else []):    # the ring's views differ: size the lists for the largest
        v_, k_ = cam_tensors(c)
        need = max(need, int(forward(v_, k_)[2]["n_isects"][0]))
    cap = int(need * 1.25) + 4096

    # ---- forward frames through the library's FrameRenderer ---------------------------------
    # One HIP graph per in-flight slot, camera in device buffers, `inflight` independent frames
    # on their own streams: the latency-bound binning kernels of one frame run under the
    # VALU-bound raster of another.  Every submit is one whole frame (projection + binning +
    # raster) and every timed region ends with a full sync.
    from robosimgs_amd import FrameRenderer, frame_to_u8
    from robosimgs_amd.dataset import frame_to_dataset
    K_host = np.asarray(sizing_cam.K, dtype=np.float64)        # the ring's cameras share their intrinsics
    n_fl = max(1, a.inflight)
    # FrameRenderer keeps its own copy of the resident scene in Morton order of the means (a one-off at construction);
    # the same frames with the scene in the order it was given are timed further down and reported beside `value`
    fr = FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=n_fl, isect_capacity=cap,
                       reorder=None if a.no_reorder else "morton")
    # N > 1 with a dataset payload: the raster writes the payload itself (RGBA8 + ray distance from its epilogue: 6 - 8 B per
    # pixel leave the kernel instead of 20, no conversion pass) and the frame is copied into the gather batch on the slot's
    # own stream; --convert-on-consumer keeps the float frames and the consumer-side conversion
    do_gather = use_dist and not a.no_gather
    # the legs of an N > 1 run: (name, payload, shard); the first is the one `value` is measured on
    legs = [(a.gather_dtype, a.gather_dtype, shard)]
    if do_gather and not a.one_payload:
        legs += [(m_, m_, shard) for m_ in ("fp32", "dataset", "dataset16") if m_ != a.gather_dtype]
        if shard_alt is not None:
            legs.append(("dataset16_weighted_root", "dataset16", shard_alt))
    fr_by_payload = {}
    if do_gather and not a.convert_on_consumer:
        for m_, dt_ in (("dataset16", torch.float16), ("dataset", torch.float32)):
            if any(l[1] == m_ for l in legs):
                fr_by_payload[m_] = FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=n_fl, isect_capacity=cap,
                                                  reorder=None if a.no_reorder else "morton", dataset_output=dt_,
                                                  dataset_K=K_host)
    fr_plain = fr
    vm_dev, K_dev = vm[0].contiguous(), K[0].contiguous()
    frames_per_step = len(shard["cam_devs"])           # 1 at N = 1, this rank's share of the ring otherwise

    comm_dev = "cpu" if debug_gloo else dev
    # Frames leave in batches of `gather_batch` through a double-buffered staging area: the frame is
    # converted (u8) or copied (fp32, dataset frames out of the raster) into its place in the batch, the slot is released at
    # once, and every gather_batch-th frame one collective ships the whole batch.
    per_rank = {}                                  # HIP-event split of this rank's last timed region, per leg (N > 1)

    def time_frames(g_mode, sh=shard, leg=None):
        """Warm-up, then regions of K steps until min_seconds are timed; the gathered payload is the dataset frame
        (RGBA8 plane + ray-distance plane), fp32 RGB + expected depth + alpha, or the 8-bit RGB image; `sh` says which
        cameras this rank renders (make_shard).  Returns (regions, collectives)."""
        nonlocal fr
        fr = fr_by_payload.get(g_mode, fr_plain) if ring else fr
        cam_devs, GB, weights = sh["cam_devs"], sh["GB"], sh["weights"]
        leg = leg or g_mode
        g_u8 = g_mode == "u8"
        g_ds = g_mode in ("dataset", "dataset16")
        g_dist = torch.float16 if g_mode == "dataset16" else torch.float32
        g_dtype = torch.float32 if g_mode == "fp32" else torch.uint8
        g_ch = {"dataset16": 6, "dataset": 8, "u8": 3, "fp32": 5}[g_mode]  # dataset: per frame [RGBA8 plane | distance plane] as bytes
        ev = {"start": None, "render_done": None, "convert": [], "last": None, "end": None}      # timed events of the region
        batch_shape = (GB, H, W, g_ch)
        staging = [torch.empty(batch_shape, device=dev, dtype=g_dtype) for _ in range(2)] if do_gather else None
        host_staging = ([torch.empty(batch_shape, device="cpu", dtype=g_dtype) for _ in range(2)]
                        if do_gather and debug_gloo else None)          # gloo debugging mode only
        gather_bufs = None
        if do_gather and rank == 0:
            gather_bufs = [[torch.empty(batch_shape, device=comm_dev, dtype=g_dtype) for _ in range(world)]
                           for _ in range(2)]
        pending = [None, None]
        last_work = [None, None]
        state = {"cur": 0, "fill": 0, "shipped": 0, "target": 0}
        sizes_all = sh["sizes"]
        tickets = []

        def ship():
            """One collective for the frames staged so far (stream-ordered after their conversion)."""
            cur = state["cur"]
            src = staging[cur]
            if debug_gloo:
                host_staging[cur].copy_(src)                           # synchronous host copy
                src = host_staging[cur]
            pending[cur] = dist.gather(src, gather_bufs[cur] if rank == 0 else None, dst=0, async_op=True)
            last_work[cur] = pending[cur]                  # (the submit side's copies into this buffer wait for it)
            state["cur"], state["fill"] = cur ^ 1, 0
            state["shipped"] += 1

        def retire():
            """Fetch the oldest frame; with N > 1 stage it for the (asynchronous) RCCL gather."""
            tk = tickets.pop(0)
            f = fr.fetch(tk, check=False)
            c0 = None
            if do_gather and ev["start"] is not None:
                c0 = torch.cuda.Event(enable_timing=True)
                c0.record()                                # (behind the wait for the slot: the conversion alone is timed)
            if do_gather:
                cur, j = state["cur"], state["fill"]
                if j == 0 and pending[cur] is not None:    # the collective that last read this staging buffer
                    pending[cur].wait()                    # (NCCL: the current STREAM waits, not the host)
                    pending[cur] = None
                if g_ds and "dataset" in f:                # written by the raster and copied into the batch on the slot's
                    torch.cuda.current_stream().wait_event(copied.pop(0))     # stream (submit): only wait for it
                elif g_ds:                                 # RGBA8 + ray distance, one kernel, inside the timed region
                    flat = staging[cur][j].view(-1)
                    frame_to_dataset(f["colors"], f["alphas"], K_host, out=(flat[:H * W * 4].view(H, W, 4),
                                                                            flat[H * W * 4:].view(g_dist).view(H, W, 1)))
                elif g_u8:                                 # quantise on the device, inside the timed region
                    frame_to_u8(f["colors"], f["alphas"], out=staging[cur][j].view(-1, 3))
                else:                                      # RGB + depth | alpha, straight from the slot's buffers
                    staging[cur][j][..., :4].copy_(f["colors"], non_blocking=True)
                    staging[cur][j][..., 4:].copy_(f["alphas"], non_blocking=True)
                state["fill"] = j + 1
                if c0 is not None:
                    c1 = torch.cuda.Event(enable_timing=True)
                    c1.record()
                    ev["convert"].append((c0, c1))
                    ev["last"] = c1
                if state["fill"] == GB:
                    state.update(last_buf=cur, last_fill=GB)
                    ship()
            fr.release(tk)

        in_graph = do_gather and g_ds and fr.dataset_dtype is not None and GB >= n_fl - 1
        copied, sub = [], {"q": 0, "cur0": 0}

        def submit(cam_dev):
            if len(tickets) == n_fl:
                retire()
            tickets.append(fr.submit(cam_dev))
            if in_graph:
                # the frame's place in its batch is known now (frames retire in the order they were submitted): the copy
                # into the batch rides on the SLOT's stream behind the frame -- the consumer stream issues no kernel at
                # all (a fourth stream of work beside three frames costs this loop 8 %)
                slot = fr._slots[tickets[-1]]
                cur, j = (sub["cur0"] + sub["q"] // GB) & 1, sub["q"] % GB
                with torch.cuda.stream(slot["stream"]):
                    if last_work[cur] is not None:            # the collective that last read this staging buffer (issued by
                        last_work[cur].wait()                 # now: GB >= frames in flight - 1): EVERY slot's stream waits
                    staging[cur][j].view(-1).copy_(slot["ds"]["dataset"], non_blocking=True)
                    e_ = torch.cuda.Event()
                    e_.record()
                copied.append(e_)
                sub["q"] += 1
            if ev["start"] is not None:                    # when this rank's renders are done: an event behind the frame
                d = torch.cuda.Event(enable_timing=True)
                d.record(fr._slots[tickets[-1]]["stream"])
                ev["render_done"] = d

        def step():
            for cd in cam_devs:
                submit(cd)

        def drain():
            while tickets:
                retire()
            if do_gather and state["fill"] > 0:            # a partial last batch still travels (whole buffer)
                state.update(last_buf=state["cur"], last_fill=state["fill"])
                ship()
            if do_gather and world > 1:
                # unequal shards (weighted, or 64 cameras over e.g. 3 ranks): ranks with fewer frames issue padding
                # collectives until every rank has made the same number of gather calls in this region.  The count is
                # arithmetic every rank does alike -- no collective decides it, so the ranks' sequences of
                # collectives cannot get out of step (a gather on one rank never meets an all-reduce on another)
                while state["shipped"] < state["target"]:
                    ship()
            for k in range(2):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None
            sub.update(q=0, cur0=state["cur"])             # (every batch shipped: the next frame opens a batch)

        def region(k_steps):
            barrier_sync(use_dist)
            # gather calls every rank will have made when this region ends: the largest shard's batches
            state["target"] = state["shipped"] + max(math.ceil(sz * k_steps / GB) for sz in sizes_all)
            t0 = time.perf_counter()
            if do_gather:
                ev.update(start=torch.cuda.Event(enable_timing=True), render_done=None, convert=[], last=None)
                ev["start"].record()
            for _ in range(k_steps):
                step()
            drain()
            if do_gather:
                ev["end"] = torch.cuda.Event(enable_timing=True)
                ev["end"].record()
            barrier_sync(use_dist)
            if do_gather:          # this rank's split of the region (HIP events): renders, conversions, the tail behind them
                s0 = ev["start"]
                per_rank[leg] = {
                    "frames": len(cam_devs) * k_steps,
                    "render_span_ms": round(s0.elapsed_time(ev["render_done"]), 3) if ev["render_done"] is not None else 0.0,
                    "convert_ms": round(sum(x.elapsed_time(y) for x, y in ev["convert"]), 3),
                    "after_last_convert_ms": round(ev["last"].elapsed_time(ev["end"]), 3) if ev["last"] is not None
                    else round(s0.elapsed_time(ev["end"]), 3),
                    "region_ms": round(s0.elapsed_time(ev["end"]), 3)}
            dt = time.perf_counter() - t0
            if use_dist:                                   # MAX over ranks; every rank sees the same number
                tt = torch.tensor([dt], device=comm_dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            return dt

        state["target"] = max(math.ceil(sz * a.warmup / GB) for sz in sizes_all) if do_gather else 0
        for _ in range(a.warmup):
            step()
        drain()
        regs = []
        while True:                                        # identical trip count on every rank (dt is all-reduced)
            regs.append(region(a.steps))
            if sum(regs) >= a.min_seconds or len(regs) >= 400:
                break
        if do_gather and rank == 0:
            # the collective really delivered every rank's frame (all cameras see the scene)
            for r_ in range(world):
                assert float(gather_bufs[0][r_][0].float().abs().max()) > 0.0, f"rank {r_}: empty gathered frame"
        if do_gather and g_ds and cam_devs and "last_buf" in state:
            # every rank, outside the timed regions, no collective: the batch it shipped last holds, frame by frame and byte
            # for byte, the dataset frames of the cameras that were due -- the loop's bookkeeping (which frame goes where in
            # which buffer, who waits for whom) checked against frames rendered afresh
            torch.cuda.synchronize()
            total, nl, b = len(cam_devs) * a.steps, state["last_fill"], state["last_buf"]
            for i in range(nl):
                tk = fr.submit(cam_devs[(total - nl + i) % len(cam_devs)])
                f = fr.fetch(tk, check=False)
                if "dataset" in f:
                    exp = f["dataset"]
                else:
                    exp = torch.empty_like(staging[b][i]).view(-1)
                    frame_to_dataset(f["colors"], f["alphas"], K_host, out=(exp[:H * W * 4].view(H, W, 4),
                                                                            exp[H * W * 4:].view(g_dist).view(H, W, 1)))
                torch.cuda.synchronize()
                assert torch.equal(staging[b][i].view(-1), exp.view(-1)), f"rank {rank}: frame {i} of the last shipped batch is not the camera's frame"
                fr.release(tk)
        return regs, state["shipped"]

    g_mode = a.gather_dtype
    g_u8 = g_mode == "u8"
    regions, n_collectives = time_frames(g_mode)
    elapsed = float(np.median(regions))
    # the named alternates, timed the same way in the same run (N > 1 only): smaller payloads, and the weighted shard
    alternates = {}
    for leg_name, leg_mode, leg_shard in legs[1:]:
        o_regions, _ = time_frames(leg_mode, leg_shard, leg_name)
        o_el = float(np.median(o_regions))
        alternates[leg_name] = {"payload": PAYLOADS[leg_mode], "bytes_per_frame": PAYLOAD_BYTES_PER_PX[leg_mode] * W * H,
                                "frames_per_rank": leg_shard["sizes"], "root_weight": leg_shard["root_weight"],
                                "frames_per_s": round(RING * a.steps / o_el, 2), "ms_per_step": round(o_el / a.steps * 1e3, 4),
                                "within_1e-4_at_the_root": leg_mode == "fp32"}
    # the same frames with the renderer keeping the caller's order (N = 1 only; half the timed span)
    given_order = None
    if not ring and not a.no_reorder:
        fr_main, keep = fr, a.min_seconds
        fr = FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=n_fl, isect_capacity=cap, reorder=None)
        a.min_seconds = keep / 2
        g_regions, _ = time_frames(g_mode)
        a.min_seconds = keep
        given_order = float(np.median(g_regions))
        status_given = fr.isect_status_max()
        assert status_given == 0
        del fr
        fr = fr_main
    # "make the radius rule a compile-time policy so the tighter one can be benchmarked" -- the same frames
    # with gsplat >= 1.5's per-axis opacity-aware extents instead of A.2 step 5's radius (N = 1 only; a quarter of the span).
    # NOT the headline: the rule changes edge pixels, the parity claim is made for the classic rule.
    rule_leg = None
    if not ring and not a.no_reorder:
        # (the classic rule is timed again beside it, the two renderers built back to back and taking turns;
        #  scripts/dbg/radius_rule_ab.py does the same over all four rule x bounds combinations)
        fr_main, keep = fr, a.min_seconds
        n_classic = max(int(s_["meta"]["n_isects"].max().item()) for s_ in fr._slots)
        pair = {"opacity_aware": FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=n_fl, isect_capacity=cap,
                                               radius_rule="opacity_aware"),
                "classic": FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=n_fl, isect_capacity=cap)}
        a.min_seconds = keep / 8
        rates = {k: [] for k in pair}
        for _ in range(2):
            for k in pair:
                fr = pair[k]
                r_regions, _ = time_frames(g_mode)
                rates[k].append(a.steps / float(np.median(r_regions)))
        a.min_seconds = keep
        assert max(f_.isect_status_max() for f_ in pair.values()) == 0
        rule_leg = {"frames_per_s": round(max(rates["opacity_aware"]), 2),
                    "frames_per_s_classic_rule_same_harness": round(max(rates["classic"]), 2),
                    "n_isect": max(int(s_["meta"]["n_isects"].max().item()) for s_ in pair["opacity_aware"]._slots),
                    "n_isect_classic_rule_tight_rectangles": n_classic, "n_isect_classic_rule_classic_rectangles": n_isect,
                    "note": "rasterization(radius_rule='opacity_aware'): extents min(3.33, sqrt(2 ln(255 o))) sqrt(Sigma_ii) per axis "
                            "(gsplat >= 1.5); with opacities the box is the bounding box of the alpha >= 1/255 ellipse, i.e. what "
                            "the tightened rectangles of the classic rule already cut the lists down to, plus the pairs the "
                            "classic square never had (beyond 3 sigma of opaque Gaussians); two renderers taking turns, best of "
                            "two short regions each"}
        del pair
        fr = fr_main
    # single-frame latency (one slot, nothing else in flight), for reference
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        tk = fr.submit(vm_dev, K_dev)
        fr.fetch(tk, check=False)
        fr.release(tk)
        torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t1) / 20 * 1e3
    status = fr.isect_status_max()
    assert status == 0, "tile-intersection capacity overflow inside the timed region"
    total_frames = (RING if ring else 1) * a.steps     # all ranks together, per region
    frames_per_s = total_frames / elapsed
    ms_per_step = elapsed / a.steps * 1e3

    if ring:
        workload = (f"configs[3]: {a.n} Gaussians, SH degree {deg}, {RING} novel-view cameras {W}x{H} "
                    f"(theta_k = 2 pi k / {RING}), sharded {'/'.join(str(x) for x in shard['sizes'])} views over "
                    f"{world} GPU(s), RCCL gather of the {g_mode} frames to rank 0; one step = one pass over the ring")
    else:
        workload = (f"configs[1]: {a.n} Gaussians, SH degree {deg}, {W}x{H} forward render "
                    f"(render_mode {MODE}: RGB + expected depth + alpha), one camera per step")
    nccl_ver = None
    if use_dist and not debug_gloo:
        try:
            nccl_ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            nccl_ver = "unknown"
        assert dist.get_backend() == "nccl" and nccl_ver, "an N > 1 run on GPUs goes over RCCL (backend nccl) and says which"
    # the ring's prediction for N = 2 / 4 / 8 from this run's one-rank rates (per rank: frames of this rank per second)
    one_rank = {}
    if not ring:
        one_rank["fp32"] = frames_per_s
    elif world == 1:
        one_rank["fp32"] = frames_per_s if g_mode == "fp32" else None
        for leg_name, leg in (alternates or {}).items():
            if leg_name in PAYLOAD_BYTES_PER_PX:
                one_rank[leg_name] = leg["frames_per_s"]
    else:                                           # N ranks measured: per-rank shares of the measured legs
        one_rank["fp32"] = frames_per_s / world if g_mode == "fp32" else None
    one_rank = {k: v for k, v in one_rank.items() if v}
    all_ranks = None
    if do_gather:                                  # every rank's split of its last region, by payload
        all_ranks = [None] * world
        dist.all_gather_object(all_ranks, per_rank)
    result = {
        "metric": "frames/sec + ms/frame (fwd, fwd+bwd) at 1M Gaussians 1920x1080",
        "value": round(frames_per_s, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong" if ring else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "render_mode": MODE,
                   "n_gaussians": a.n, "n_visible": n_vis, "n_isect": n_isect,
                   "n_isect_binned": n_isect_binned,
                   "tiles": tile_w * tile_h,
                   "world_size": dist.get_world_size() if use_dist else 1, "rccl_version": nccl_ver,
                   "frames_per_step_all_ranks": RING if ring else 1,
                   "frames_per_step_this_rank": frames_per_step,
                   "frames_per_rank": shard["sizes"],
                   "root_weight": shard["root_weight"],
                   "per_rank": ({m: [r_.get(m) for r_ in all_ranks] for m in all_ranks[0]} if all_ranks else None),
                   "per_rank_note": ("HIP events on each rank over its last timed region: render_span = region start -> the rank's "
                                     "last frame rendered, convert = sum of the payload conversions, after_last_convert = what "
                                     "the rank's stream still waited for (its collectives; on rank 0 the other ranks' frames)"
                                     if all_ranks else None),
                   "root_bound": ({m: {"bytes_per_frame": PAYLOAD_BYTES_PER_PX[m] * W * H,
                                       "frames_per_s_ceiling": round(root_bound_frames_per_s(m, W, H, world), 0)}
                                   for m in PAYLOADS} if (ring and world > 1) else None),
                   "root_bound_note": (f"rank 0's inbound xGMI at {XGMI_LINK_GBS:g} GB/s per link (assumed, not measured) carries "
                                       "(N - 1) / N of every frame: the ceiling of a single-root gather whatever the renderers do"
                                       if (ring and world > 1) else None),
                   "scaling_prediction": scaling_prediction(one_rank, W, H) if one_rank.get("fp32") else None,
                   "scaling_prediction_note": ("min(N x this run's one-rank rate, the single root's ceiling) per payload at N = 2 / 4 / "
                                               "8: what the driver's SCALE record can be read against; the root's receive load on its "
                                               f"own rendering and the assumed {XGMI_LINK_GBS:g} GB/s per link are not in it"),
                   "gather": ((PAYLOADS[g_mode]
                               + f" to rank 0 (RCCL), {shard['GB']} frames per collective, "
                                 f"{n_collectives} collectives issued") if do_gather else "none"),
                   "alternates": alternates or None,
                   "alternates_note": ("the same ring in the same run with a smaller payload and / or a smaller block on the "
                                       f"gathering rank; `value` is the {g_mode} leg on blocks of {'/'.join(str(x) for x in shard['sizes'])} (configs[3] as written: equal blocks, fp32 frames). With "
                                       "fp32 frames a single root cannot reach 6 x the one-GPU rate (root_bound): a 6 x claim can "
                                       "only be made on a dataset leg, and says so" if alternates else None),
                   "launch": f"one HIP graph per frame, no host read-back, {n_fl} independent "
                             "frames in flight on separate HIP streams",
                   "frames_in_flight": n_fl, "single_frame_latency_ms": round(latency_ms, 4),
                   "scene_order": ("as given" if a.no_reorder else
                                   "FrameRenderer's own copy in Morton order of the means (sorted once at construction; same "
                                   "image except where two Gaussians of a pixel tie in depth to the last bit)"),
                   "frames_per_s_scene_in_given_order": (round(total_frames / given_order, 2) if given_order else None),
                   "radius_rule_opacity_aware": rule_leg,
                   "timing": f"median of {len(regions)} regions of {a.steps} steps, each bracketed by "
                             f"barrier + synchronize, MAX over ranks (min {min(regions) * 1e3:.2f} ms, "
                             f"max {max(regions) * 1e3:.2f} ms per region)"},
    }

    if rank == 0:
        ch = 4
        t_given, t = t, fr.t        # the kernels below are timed on the scene as the timed frames hold it
        # ---- roofline of the dominant kernel (tile raster forward), HIP events on the stream
        radii, m2d, depths, con, _, feats, splats = ops.project_color_fwd_raw(
            t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm[0], K[0], W,
            H, 0.3, 0.01, 1e10, 0.0, False, True, want_splats=True)
        tl = ops.isect_tiles_raw(m2d, radii, depths, tile_w, tile_h, cap, want_tiles_per_gauss=False,
                                 conics=con, opacities=t["opacities"])
        # the inference variant (no last_ids), 4 channels, expected-depth epilogue: the kernel the
        # timed frames above run
        out = None
        reps = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        # two schedules of the same blend (identical pixels): one wave per tile -- fewest instructions, what
        # the frames in flight above run -- and one wave per 8x8 block (MGS_RASTER_LATENCY), the choice for a
        # launch that has the GPU to itself.  The roofline line is the kernel the timed frames ran.
        timed_latency = n_fl == 1
        raster_times = {}
        for lat in (False, True):
            def raster():
                return ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tile_w, tile_h,
                                             tl.tile_offsets, tl.flatten_ids, out=out, track_last=False,
                                             splats=splats, expected_last=True, latency=lat,
                                             group_order=tl.group_order)
            for _ in range(5):
                out = raster()
            e0.record()
            for _ in range(reps):
                raster()
            e1.record()
            torch.cuda.synchronize()
            raster_times[lat] = e0.elapsed_time(e1) / reps
        raster_ms = raster_times[timed_latency]
        n_px = W * H
        # n_isect * 44 (id 4 + mean 8 + conic 12 + opacity 4 + rgb 12 + depth 4)
        #                 + n_px * 24 (rgb 12 + depth 4 + alpha 4 + last_id 4) + tiles * 8.
        # The contract figure prices the classic lists; the kernel walks the tightened ones and the
        # inference variant does not store last_ids, so the bytes it really needs are fewer: both fracs.
        algo_bytes = n_isect * 44 + n_px * 24 + tile_w * tile_h * 8
        walked_bytes = n_isect_binned * 44 + n_px * 20 + tile_w * tile_h * 8
        achieved = algo_bytes / (raster_ms * 1e-3) / 1e9
        achieved_walked = walked_bytes / (raster_ms * 1e-3) / 1e9
        traffic, traffic_note = pmc_traffic("raster_fwd_q" if timed_latency else "raster_fwd", (a.n, W, H, deg) == (1_000_000, 1920, 1080, 3))
        result["roofline"] = {"kernel": (f"raster_fwd_q_kernel<{ch}, false, false>" if timed_latency
                                         else f"raster_fwd_kernel<{ch}, false, false>"), "bound": "hbm",
                              "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 4),
                              "traffic": traffic, "traffic_source": traffic_note,
                              "algorithmic_bytes": algo_bytes,
                              "kernel_ms": round(raster_ms, 4),
                              "kernel_ms_by_schedule": {"throughput (one wave per tile)": round(raster_times[False], 4),
                                                        "latency (one wave per 8x8 block)": round(raster_times[True], 4)},
                              "on_walked_lists": {"bytes": walked_bytes, "achieved": round(achieved_walked, 1),
                                                  "frac": round(achieved_walked / HBM_PEAK_GBS, 4),
                                                  "note": "n_isect_binned * 44 + n_px * 20 + tiles * 8: the "
                                                          "tightened lists the kernel walks, no last_ids store"},
                              "valu": pmc_valu("raster_fwd_q_kernel<4, false, false>" if timed_latency else "raster_fwd_kernel<4, false, false>",
                                               (a.n, W, H, deg) == (1_000_000, 1920, 1080, 3)),
                              "note": "VALU-bound kernel (DESIGN.md 4.3): `valu` says how close it runs to the issue rate "
                                      "of its own instruction mix; the HBM fraction is reported as the contract asks"}

        # the second-largest forward kernel is HBM-bound: projection + SH colour (SURVEY.md 8(d):
        # N*236 + n_vis*48 algorithmic bytes), timed the same way
        def proj():
            return ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg,
                                             t["colors"], vm[0], K[0], W, H, 0.3, 0.01, 1e10, 0.0, False,
                                             True, want_splats=True, bin_seed="tight", lean=True)
        for _ in range(5):
            proj()
        e0.record()
        for _ in range(reps):
            proj()
        e1.record()
        torch.cuda.synchronize()
        proj_ms = e0.elapsed_time(e1) / reps
        proj_bytes = a.n * (44 + 12 * (deg + 1) ** 2) + n_vis * 48
        result["roofline_projection"] = {
            "kernel": f"project_color_fwd_kernel<{deg}, {'true' if deg >= 2 else 'false'}>", "bound": "hbm",
            "achieved": round(proj_bytes / (proj_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(proj_bytes / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes": proj_bytes, "kernel_ms": round(proj_ms, 4),
            "note": "the inference-frame form the timed frames run: writes depths, the 48-byte splat records and the "
                    "binning seed (n_vis * 48 + N * 8 more bytes, not counted); radii / means2d / conics / feats are "
                    "NULL (nobody reads them in such a frame)"}

        # binning stage (rectangles .. depth-ordered lists + tile offsets) as one HIP-event interval; SURVEY.md 8(d):
        # n_vis*20 + n_isect*12 (count + emit) + n_isect*24 (sort, ideal 1R+1W) + n_isect*8 + tiles*4.
        # Timed the way the frame graph runs it -- seeded with the rectangles and counts the fused projection
        # kernel wrote -- and as the standalone operator, which computes them itself first.
        seed = proj()[-1]
        def binning(seeded):
            # the seed's per-64 sums are consumed by the call (scanned in place on the paths that need them): every
            # call gets a fresh copy of the 62 KB array, as every frame gets fresh sums from its projection kernel
            return ops.isect_tiles_raw(m2d, radii, depths, tile_w, tile_h, cap, want_tiles_per_gauss=False,
                                       conics=con, opacities=t["opacities"],
                                       seed=(seed[0], seed[1].clone()) if seeded else None, want_tile_ids=not seeded)
        bin_times = {}
        for seeded in (True, False):
            for _ in range(5):
                tl_b = binning(seeded)
            e0.record()
            for _ in range(reps):
                binning(seeded)
            e1.record()
            torch.cuda.synchronize()
            bin_times[seeded] = e0.elapsed_time(e1) / reps
            assert int(tl_b.n_isect.item()) == n_isect_binned, "seeded and unseeded binning disagree"
        bin_ms = bin_times[True]
        bin_bytes = n_vis * 20 + n_isect * 44 + tile_w * tile_h * 4
        result["roofline_binning"] = {
            "kernels": "mgs_isect_tiles: histogram of the tile groups, column scan, scatter, per-tile depth sort "
                       "(+ tile rectangles and counts when not seeded by the projection kernel)", "bound": "hbm",
            "achieved": round(bin_bytes / (bin_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(bin_bytes / (bin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes": bin_bytes, "stage_ms": round(bin_ms, 4),
            "stage_ms_standalone_operator": round(bin_times[False], 4),
            "note": "eager launches back to back on one stream (the frame graph replays the same kernels); "
                    "algorithmic bytes price the classic lists (n_isect), the stage bins the tightened ones"}

        # ---- training-step variant (configs[2]): forward + L1 + backward ------------------
        std = (a.n, W, H, deg) == (1_000_000, 1920, 1080, 3)
        try:
            result["fwd_bwd"] = bench_fwd_bwd(a, t_given, vm, K, W, H, deg, cap, dev)      # a trainer's own order
            result["fwd_bwd"]["scene_order"] = "as given"
            result["fwd_bwd"]["roofline"] = bwd_roofline(t_given, vm, K, W, H, deg, cap, n_isect, std)
            # the same step with the parameters in Morton order of the means (what a trainer gets by re-ordering its
            # parameter tensors and optimiser state with robosimgs_amd.pipeline.locality_order every few hundred steps)
            if t is not t_given:
                m = bench_fwd_bwd(a, t, vm, K, W, H, deg, cap, dev)
                result["fwd_bwd"]["morton_order"] = {"ms_per_step": m["ms_per_step"], "steps_per_s": m["steps_per_s"]}
        except Exception as e:  # keep the headline line even if this leg fails
            result["fwd_bwd"] = {"error": repr(e)[:200]}

        # ---- configs[4]: 5 M Gaussians at 3840x2160 (HBM-pressure / tile-overflow stress), per stage ----------
        if not a.no_stress and not ring:          # (N = 1 only: in a multi-GPU run the other ranks would wait for it)
            try:
                del radii, m2d, depths, con, feats, splats, tl, tl_b, out, seed
                torch.cuda.empty_cache()
                result["stress_4k"] = stress_4k(dev, deg, n_fl)
            except Exception as e:
                result["stress_4k"] = {"error": repr(e)[:200]}

        # ---- a scene shaped like an export (not a BASELINE config): no cliff where real scenes live ----------
        if not a.no_stress and not ring:
            try:
                torch.cuda.empty_cache()
                result["heavy_tailed"] = heavy_tailed_leg(a, dev, deg, n_fl)
            except Exception as e:
                result["heavy_tailed"] = {"error": repr(e)[:200]}

        # ---- CPU baseline: the oracle's C++/OpenMP port on this box's host cores -----------
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene, sizing_cam, W, H, deg, a.cpu_seconds)
        if out_fd is None:
            print(json.dumps(result), flush=True)
        else:
            os.write(out_fd, (json.dumps(result) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
        if world > 1:
            # the ranks leave without the interpreter's teardown: everything is printed and the group is gone, and the
            # teardown of HIP graphs / streams / communicators beside the other ranks' is the one place a finished run
            # could still get stuck (a two-rank run was once seen not to return; not reproduced in eight repeats)
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)


def pmc_traffic(kernel_key, standard_workload):
    """HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (scripts/pmc_traffic.sh
    writes profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 correction
    applied).  Counter passes cannot run inside bench.py, so the number is only printed when it was
    taken from THIS build of libmgs.so (the file records the library's build stamp) on this workload."""
    if not standard_workload:
        return None, "not the configs[1] workload: no PMC measurement applies"
    try:
        with open(PMC_FILE) as f:
            rec = json.load(f)
        from robosimgs_amd.csrc import build as hip_build
        stamp = hip_build.current_stamp()
        k = rec["kernels"][kernel_key]
        if rec.get("stamp") != stamp:
            return None, (f"profiles/pmc_traffic.json was measured on build {str(rec.get('stamp'))[:12]}, this is "
                          f"{stamp[:12]}: stale, not printed (re-run scripts/pmc_traffic.sh)")
        return int(k["traffic_bytes"]), k.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE")
    except Exception as e:
        return None, f"no PMC record ({type(e).__name__})"


# Issue cost of the blend's instruction mix, SIMD-cycles per wave64 vector instruction (scripts/ubench/valu_issue.hip,
# DESIGN.md 4.0): the shipped lane-mask body (raster_fwd.hip: blend_pixel_safe_asm) is 12 fma-class at 2.4, 2 compares
# at 4.1 and one v_exp at 8.15 = 45 cycles per 15 instructions
VALU_ISSUE_FLOOR = 45.0 / 15.0
# the backward's quadrant body (raster_bwd.hip: grad_pixel, SAFE): 24 fma-class, 2 compares + 1 select, v_exp + v_rcp
VALU_ISSUE_FLOOR_BWD = (24 * 2.4 + 3 * 4.1 + 2 * 8.15) / 29.0


def pmc_valu(kernel_name, standard_workload, stage_prefix="raster_inf", floor=None):
    """What bounds the raster for real: vector instructions per launch and SIMD-cycles per instruction from the
    same PMC record (SQ_INSTS_VALU; GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 x 1024 SIMDs), against the issue
    cost of the blend's instruction mix.  Printed under the same build-stamp rule as `traffic`."""
    if not standard_workload:
        return None
    try:
        with open(PMC_FILE) as f:
            rec = json.load(f)
        from robosimgs_amd.csrc import build as hip_build
        if rec.get("stamp") != hip_build.current_stamp():
            return None
        row = next(v for k, v in rec["raw"].items() if kernel_name in k and k.startswith(stage_prefix))
        insts, gui = float(row["SQ_INSTS_VALU"]), float(row["GRBM_GUI_ACTIVE"])
        cyc = gui / 8.0 * 1024.0 / insts
        floor = VALU_ISSUE_FLOOR if floor is None else floor
        return {"bound": "valu-issue", "instructions_per_launch": int(insts),
                "simd_cycles_per_instruction": round(cyc, 2),
                "issue_floor_cycles_per_instruction": round(floor, 2),
                "frac_of_issue_bound": round(floor / cyc, 3),
                "note": "the floor is the blend loop's mix; fetch, cull and queue instructions are mostly of the cheaper "
                        "class, so a kernel at the bound can read slightly above 1",
                "source": "rocprofv3 --pmc SQ_INSTS_VALU, GRBM_GUI_ACTIVE in the run that took `traffic` "
                          "(profiles/pmc_traffic.json, profiles/r5/06_pmc_counters.md); floor: "
                          "scripts/ubench/valu_issue.hip"}
    except Exception:
        return None


def bench_fwd_bwd(a, t, vm, K, W, H, deg, cap, dev):
    from robosimgs_amd import l1_loss, unit_gradient
    names = ("means", "quats", "scales", "opacities", "colors")
    params = {k: t[k].detach().clone().requires_grad_(True) for k in names}
    # L1 to a U(0,1) target (seed 1) on all four channels of the RGB+ED frame
    target = torch.rand(1, H, W, 4, device=dev, generator=torch.Generator(dev).manual_seed(1))

    def train_step():
        for p in params.values():
            p.grad = None
        colors, alphas, meta = rasterization(params["means"], params["quats"], params["scales"],
                                             params["opacities"], params["colors"], vm, K, W, H,
                                             sh_degree=deg, render_mode=MODE, isect_capacity=cap)
        loss = l1_loss(colors, target)      # fused HIP L1 (== (colors - target).abs().mean())
        loss.backward(gradient=unit_gradient(loss))      # what robosimgs_amd.Trainer.step does: no ones_like fill, no scale launch
        return loss

    for _ in range(3):
        train_step()
    torch.cuda.synchronize()
    mode = "eager"
    runner = train_step
    try:
        side = torch.cuda.Stream(dev)
        with torch.cuda.stream(side):
            train_step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                train_step()
        torch.cuda.synchronize()
        runner, mode = g.replay, "hip-graph"
    except Exception:
        torch.cuda.synchronize()
    for _ in range(3):
        runner()
    torch.cuda.synchronize()
    times = []
    while sum(times) < 0.5 and len(times) < 50:
        t0 = time.perf_counter()
        for _ in range(a.bwd_steps):
            runner()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times)) / a.bwd_steps
    return {"workload": f"configs[2]: forward ({MODE}) + L1 loss to U(0,1) target (seed 1) + backward",
            "ms_per_step": round(dt * 1e3, 4), "steps_per_s": round(1.0 / dt, 2),
            "steps": a.bwd_steps, "regions": len(times), "launch": mode}


def bwd_roofline(t, vm, K, W, H, deg, cap, n_isect, standard_workload, segment=256, reps=30):
    """Roofline of the training step's dominant kernel, raster_bwd_kernel<4, ...> in the form the step runs it (the
    segmented walk from the forward's checkpoints, 4 channels, "ED" cotangent), timed with HIP events on the stream:
    mgs_rasterize_bwd_det(MGS_RASTER_BWD_RECORDS_ONLY) = the flag memset (4.7 MB), the unit table (one thread per tile)
    and the raster kernel -- no reduce.  Algorithmic bytes: SURVEY.md 8(d), n_isect * (44 + 36) + n_px * (24 + 20)."""
    tile_w, tile_h = -(-W // 16), -(-H // 16)
    radii, m2d, depths, con, _, feats, splats, seed = ops.project_color_fwd_raw(
        t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm[0], K[0], W, H, 0.3, 0.01, 1e10, 0.0,
        False, True, want_splats=True, bin_seed="tight")
    tl = ops.isect_tiles_raw(m2d, radii, depths, tile_w, tile_h, cap, want_tiles_per_gauss=False, want_pair_info=True,
                             seed=seed, splats=splats)
    ck = ops.checkpoint_buffer(cap, tile_w, tile_h, 4, segment, m2d.device)
    render, alphas, last = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tile_w, tile_h, tl.tile_offsets,
                                                 tl.flatten_ids, splats=splats, expected_last=True, latency=True,
                                                 group_order=tl.group_order, channels=4, checkpoints=ck,
                                                 checkpoint_interval=segment)
    gen = torch.Generator(m2d.device).manual_seed(1)
    v_r = torch.sign(render - torch.rand(H, W, 4, device=m2d.device, generator=gen)) / float(render.numel())   # the L1 cotangent
    def run(only):
        return ops.rasterize_bwd_det_raw(m2d, con, feats, t["opacities"], None, W, H, tile_w, tile_h, tl, alphas, last,
                                         v_r, None, splats=splats, expected_render=render, render_out=render,
                                         checkpoints=ck, checkpoint_interval=segment, records_only=only)

    # (1) inside a step-like sequence: every iteration runs the step's forward stages first (projection, binning, raster
    # forward with checkpoints), then the backward call between two events.  This is the duration the kernel has in the
    # training step -- what rocprofv3 reports for it over this run -- and what `achieved` is priced on.
    pairs = []
    for it in range(4 + 20):
        p_ = ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm[0], K[0], W, H,
                                       0.3, 0.01, 1e10, 0.0, False, True, want_splats=True, bin_seed="tight")
        tl_i = ops.isect_tiles_raw(p_[1], p_[0], p_[2], tile_w, tile_h, cap, want_tiles_per_gauss=False, want_pair_info=True,
                                   seed=p_[-1], splats=p_[6])
        ops.rasterize_fwd_raw(p_[1], p_[3], p_[5], t["opacities"], None, W, H, tile_w, tile_h, tl_i.tile_offsets,
                              tl_i.flatten_ids, out=(render, alphas, last), splats=p_[6], expected_last=True, latency=True,
                              group_order=tl_i.group_order, channels=4, checkpoints=ck, checkpoint_interval=segment)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        run(True)
        ev[1].record()
        run(False)
        ev[2].record()
        if it >= 4:
            pairs.append(ev)
    torch.cuda.synchronize()
    in_step = float(np.median([e[0].elapsed_time(e[1]) for e in pairs]))
    in_step_full = float(np.median([e[1].elapsed_time(e[2]) for e in pairs]))
    # (2) sustained: the call alone, ten per HIP graph, replayed back to back -- the clock settles lower under this
    # kernel's uninterrupted vector load, so this reads ~15 % above (1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = {}
    side = torch.cuda.Stream(m2d.device)
    for only in (True, False):
        with torch.cuda.stream(side):
            for _ in range(3):
                run(only)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            per_graph = 10                      # calls per graph: a graph launch has its own start-up gap, shared by ten
            with torch.cuda.graph(gr, stream=side):
                keep = [run(only) for _ in range(per_graph)]
            gr.replay()
            e0.record(side)
            for _ in range(max(1, reps // per_graph)):
                gr.replay()
            e1.record(side)
        torch.cuda.synchronize()
        times[only] = e0.elapsed_time(e1) / (max(1, reps // per_graph) * per_graph)
        del gr, keep
    n_px = W * H
    algo = n_isect * (44 + 36) + n_px * (24 + 20)
    ms = in_step
    traffic, note = pmc_traffic("raster_bwd", standard_workload)
    return {"kernel": "raster_bwd_kernel<4, false, true, false, true> (records, segments of %d list entries)" % segment,
            "bound": "hbm", "achieved": round(algo / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": note,
            "algorithmic_bytes": algo, "kernel_ms": round(ms, 4),
            "kernel_ms_covers": "flag memset (4.7 MB) + unit table kernel (~5 us) + raster_bwd_kernel: mgs_rasterize_bwd_det with "
                                "MGS_RASTER_BWD_RECORDS_ONLY between two HIP events, behind the step's own forward stages in "
                                "every iteration (median of 20); the rocprofv3 rows under profiles/ give the kernel alone",
            "with_reduce_ms": round(in_step_full, 4),
            "sustained_loop_ms": {"records_only": round(times[True], 4), "with_reduce": round(times[False], 4),
                                  "note": "the call alone, replayed back to back as HIP graphs of ten: the clock settles lower "
                                          "under this kernel's uninterrupted vector load"},
            "valu": pmc_valu("raster_bwd_kernel<4, false, true", standard_workload, "raster_bwd_split", VALU_ISSUE_FLOOR_BWD),
            "note": "VALU-issue-bound like the forward (DESIGN.md 4.4): ~107 vector instructions per evaluated (tile, "
                    "Gaussian) pair; the HBM fraction is reported as the contract asks"}


def stress_4k(dev, deg, n_fl, frames=20):
    """BASELINE configs[4]: 5 M Gaussians (mu = ln 0.008, seed 0), SH degree 3, 3840x2160, theta = 0.3, RGB+ED.  Per-stage
    HIP-event times of the inference-frame form (median of `frames` eager frames), each against its SURVEY.md 8(d) bytes
    (classic n_isect, as BASELINE.md section 3 prices them: 4.8 GB, 0.60 ms at 8 TB/s), and frames/s through FrameRenderer."""
    from robosimgs_amd import FrameRenderer
    n, mu, W, H = 5_000_000, 0.008, 3840, 2160
    scene = synthetic_scene(n, math.log(mu), deg, seed=0)
    t = scene.to_torch(dev, deg)
    del scene
    cam = camera_ring(1, W, H, thetas=[0.3])[0]
    vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
    K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
    tw, th = -(-W // 16), -(-H // 16)
    n_px, n_tiles = W * H, tw * th

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def frame(cap, bounds, rec=None, lean=True, t=t):
        e0 = ev()
        radii, m2d, dep, con, _, feats, splats, seed = ops.project_color_fwd_raw(
            t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False,
            True, want_splats=True, bin_seed=bounds, lean=lean)
        e1 = ev()
        tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, cap, want_tiles_per_gauss=False, seed=seed, want_tile_ids=False)
        e2 = ev()
        ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids,
                              splats=splats, track_last=False, expected_last=True, latency=True, group_order=tl.group_order,
                              channels=4)
        e3 = ev()
        if rec is not None:
            rec.append((e0, e1, e2, e3))
        return tl, radii

    tl, radii = frame(48_000_000, "classic", lean=False)
    torch.cuda.synchronize()
    n_isect, n_vis = int(tl.n_isect), int((radii > 0).sum())
    assert int(tl.status) == 0
    tl, _ = frame(48_000_000, "tight")
    n_binned = int(tl.n_isect)
    lens = (tl.tile_offsets[1:] - tl.tile_offsets[:-1])
    lmax, lmean = int(lens.max()), float(lens.float().mean())
    del tl, radii
    cap = int(n_binned * 1.15) + 4096
    # throughput: the same frames through FrameRenderer (one HIP graph per slot, Morton-ordered resident copy)
    fr = FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=n_fl, isect_capacity=cap)
    cam_dev = FrameRenderer.pack_camera(vm, K)
    bytes_ = {"project": n * (44 + 12 * (deg + 1) ** 2) + n_vis * 48,
              "binning": n_vis * 20 + n_isect * 44 + n_tiles * 4,
              "raster": n_isect * 44 + n_px * 24 + n_tiles * 8}
    total_bytes = sum(bytes_.values())

    def stage_times(scene_t):
        rec = []
        for _ in range(3):
            frame(cap, "tight", t=scene_t)
        for _ in range(frames):
            frame(cap, "tight", rec, t=scene_t)
        torch.cuda.synchronize()
        ts = np.array([[x[i].elapsed_time(x[i + 1]) for i in range(3)] for x in rec])
        med = np.median(ts, 0)
        st = {}
        for i, k in enumerate(("project", "binning", "raster")):
            st[k] = {"ms": round(float(med[i]), 4), "algorithmic_bytes": bytes_[k],
                     "frac": round(bytes_[k] / (med[i] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        return st, float(np.median(ts.sum(1)))

    # the stages on the scene as the timed frames hold it (FrameRenderer's Morton-ordered copy: like the headline's stage
    # figures), and on the caller's order (what a single rasterization() call on the caller's tensors runs)
    stages, total_ms = stage_times(fr.t)
    stages_given, total_given_ms = stage_times(t)
    tickets = []

    def push():
        if len(tickets) == n_fl:
            tk = tickets.pop(0)
            fr.fetch(tk, check=False)
            fr.release(tk)
        tickets.append(fr.submit(cam_dev))
    for _ in range(6):
        push()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3 * frames):
        push()
    while tickets:
        tk = tickets.pop(0)
        fr.fetch(tk, check=False)
        fr.release(tk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (3 * frames)
    status = fr.isect_status_max()
    return {"workload": f"configs[4]: {n} Gaussians, SH degree {deg}, {W}x{H} forward render ({MODE}), theta = 0.3",
            "n_visible": n_vis, "n_isect": n_isect, "n_isect_binned": n_binned, "tiles": n_tiles,
            "list_length_mean": round(lmean, 1), "list_length_max": lmax, "stages": stages,
            "scene_order": "FrameRenderer's own copy in Morton order of the means (as in the headline); stages_scene_in_given_order: the caller's tensors",
            "stages_scene_in_given_order": stages_given, "frame_ms_eager_stages_scene_in_given_order": round(total_given_ms, 4),
            "frame_ms_eager_stages": round(total_ms, 4),
            "frame_algorithmic_bytes": total_bytes,
            "frame_frac_of_hbm_roofline": round(total_bytes / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "floor_ms_at_8TBs": round(total_bytes / 8e12 * 1e3, 4),
            "frames_per_s": round(1.0 / dt, 2), "ms_per_frame": round(dt * 1e3, 4), "frames_in_flight": n_fl,
            "frames_timed": 3 * frames, "isect_overflow": bool(status),
            "timing": f"stages: median of {frames} eager frames, HIP events between the three C-ABI calls; frames/s: "
                      f"{3 * frames} graph replays through FrameRenderer after 6 warm-up frames"}


def heavy_tailed_leg(a, dev, deg, n_fl, frames=20):
    """NOT a BASELINE.json config: the 1080p frame and the training step on a scene shaped like an export
    (robosimgs_amd.synthetic_scene_heavy_tailed: clustered means, log-normal extents of sigma 1.2, 4,000 needle-like and 6
    screen-filling Gaussians among 1 M) -- that there is no cliff where real scenes live: lists of 40 ... 39 k entries per
    tile instead of 615 everywhere.  Per-stage HIP-event times (median of `frames` eager frames on the renderer's Morton
    copy), the raster under both schedules, frames/s through FrameRenderer, and the training step as bench_fwd_bwd runs it.
    Parity at this scene: tests/test_gpu_full_size.py (forward gate and the per-row gradient budget gate vs the fp64 port)."""
    from robosimgs_amd import FrameRenderer, synthetic_scene_heavy_tailed
    n, W, H = 1_000_000, 1920, 1080
    scene = synthetic_scene_heavy_tailed(n, sh_degree=deg, seed=0)
    t = scene.to_torch(dev, deg)
    del scene
    cam = camera_ring(1, W, H, thetas=[0.3])[0]
    vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)
    K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)
    tw, th = -(-W // 16), -(-H // 16)

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def frame(cap, bounds, t_, rec=None, lean=True, latency=True):
        e0 = ev()
        radii, m2d, dep, con, _, feats, splats, seed = ops.project_color_fwd_raw(
            t_["means"], t_["quats"], t_["scales"], t_["opacities"], deg, t_["colors"], vm, K, W, H, 0.3, 0.01, 1e10, 0.0, False,
            True, want_splats=True, bin_seed=bounds, lean=lean)
        e1 = ev()
        tl = ops.isect_tiles_raw(m2d, radii, dep, tw, th, cap, want_tiles_per_gauss=False, seed=seed, want_tile_ids=False)
        e2 = ev()
        ops.rasterize_fwd_raw(m2d, con, feats, t_["opacities"], None, W, H, tw, th, tl.tile_offsets, tl.flatten_ids,
                              splats=splats, track_last=False, expected_last=True, latency=latency, group_order=tl.group_order,
                              channels=4)
        e3 = ev()
        if rec is not None:
            rec.append((e0, e1, e2, e3))
        return tl, radii

    tl, radii = frame(16_000_000, "classic", t, lean=False)
    torch.cuda.synchronize()
    n_isect, n_vis = int(tl.n_isect), int((radii > 0).sum())
    assert int(tl.status) == 0
    tl, _ = frame(16_000_000, "tight", t)
    n_binned = int(tl.n_isect)
    lens = (tl.tile_offsets[1:] - tl.tile_offsets[:-1]).float()
    q = torch.quantile(lens, torch.tensor([0.5, 0.99], device=lens.device))
    stats = {"mean": round(float(lens.mean()), 1), "median": int(q[0]), "p99": int(q[1]), "max": int(lens.max()), "min": int(lens.min())}
    del tl, radii
    cap = int(n_binned * 1.25) + 4096
    fr = FrameRenderer(t, W, H, render_mode=MODE, frames_in_flight=n_fl, isect_capacity=cap)
    cam_dev = FrameRenderer.pack_camera(vm, K)
    out = {}
    for lat, name in ((True, "latency (one wave per 8x8 block)"), (False, "throughput (one wave per tile)")):
        rec = []
        for _ in range(3):
            frame(cap, "tight", fr.t, latency=lat)
        for _ in range(frames):
            frame(cap, "tight", fr.t, rec, latency=lat)
        torch.cuda.synchronize()
        ts = np.median(np.array([[x[i].elapsed_time(x[i + 1]) for i in range(3)] for x in rec]), 0)
        out[name] = {"project_ms": round(float(ts[0]), 4), "binning_ms": round(float(ts[1]), 4), "raster_ms": round(float(ts[2]), 4)}
    tickets = []

    def push():
        if len(tickets) == n_fl:
            tk = tickets.pop(0)
            fr.fetch(tk, check=False)
            fr.release(tk)
        tickets.append(fr.submit(cam_dev))
    for _ in range(6):
        push()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5 * frames):
        push()
    while tickets:
        tk = tickets.pop(0)
        fr.fetch(tk, check=False)
        fr.release(tk)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (5 * frames)
    status = fr.isect_status_max()
    step = bench_fwd_bwd(a, t, vm[None], K[None], W, H, deg, cap, dev)
    # the same step on the renderer's Morton-ordered copy: the generator APPENDS its 4,006 large rectangles, so "as given"
    # has them contiguous in index -- a handful of workgroups of the index-partitioned kernels (tile histogram, scatter,
    # record reduce) own most of the pairs; robosimgs_amd.reorder_parameters is what a trainer does about it
    step_m = bench_fwd_bwd(a, fr.t, vm[None], K[None], W, H, deg, cap, dev)
    return {"workload": f"NOT a BASELINE config: {n} Gaussians, heavy-tailed synthetic scene (synthetic_scene_heavy_tailed, seed 0), SH degree "
                        f"{deg}, {W}x{H} ({MODE}), theta = 0.3",
            "n_visible": n_vis, "n_isect": n_isect, "n_isect_binned": n_binned, "list_length": stats,
            "stages_by_raster_schedule": out, "frames_per_s": round(1.0 / dt, 2), "ms_per_frame": round(dt * 1e3, 4),
            "frames_in_flight": n_fl, "isect_overflow": bool(status),
            "fwd_bwd": {"ms_per_step": step["ms_per_step"], "launch": step["launch"],
                        "scene_order": "as given (the large rectangles contiguous at the end of the index range)",
                        "morton_order": {"ms_per_step": step_m["ms_per_step"]}},
            "note": "the raster is a tile's serial walk: the launch lasts as long as its longest lists (the tile-time tail), "
                    "which is what the two schedules' raster_ms against the 1080p headline's show"}


def cpu_baseline(scene, cam, W, H, deg, budget_s):
    from oracle import cpu_ref
    threads = cpu_ref.max_threads()
    args = (scene.means, scene.quats, scene.scales, scene.opacities, scene.sh_coeffs,
            cam.viewmat(), cam.K, W, H, deg)
    base_lib, base_flags = cpu_ref.baseline_lib()           # the AVX2 + FMA build (x86-64-v3) where this host can run it
    kw = dict(with_depth=True, library=base_lib)            # RGB + depth sum + alpha, like the GPU frames
    for _ in range(3):                                      # BASELINE.md protocol: 3 warm-ups
        cpu_ref.render(*args, **kw)
    times = []
    t_start = time.perf_counter()
    while (time.perf_counter() - t_start) < budget_s and len(times) < 50:
        t0 = time.perf_counter()
        _, _, info = cpu_ref.render(*args, **kw)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} full frames of the same 1M-Gaussian 1080p RGB+depth+alpha workload "
                      f"after 3 warm-ups, median {med * 1e3:.1f} ms/frame; oracle/gs_cpu.cpp (fp32 "
                      f"instantiation), OpenMP, {threads} threads on {model}; built with g++ {' '.join(base_flags)} "
                      "(not -march=native: the .so is compiled in the dev container and must run on the GPU box's host)",
            "pair_evals": info["pair_evals"]}


if __name__ == "__main__":
    main()
