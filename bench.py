#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step is one forward render of the 1 M-Gaussian, SH-degree-3 synthetic scene at 1920x1080
(BASELINE.json configs[1]) per rank: projection+SH, tile binning, tile raster, all through
the C ABI of libmgs.so, replayed as one HIP graph with the scene resident in HBM.  For N > 1
(launched by torch.distributed.run, one process per GPU over RCCL) every rank renders its own
camera of the ring and rank 0 gathers the finished fp32 RGB frames (config 4's collective);
value = frames all ranks rendered / max-over-ranks time, so scaling is weak.

Rank 0 prints ONE JSON line.  Beside the contract fields it carries
  roofline      tile-raster forward kernel: algorithmic bytes / HIP-event time vs 8 TB/s
  cpu_baseline  the C++/OpenMP port in oracle/gs_cpu.cpp timed on this box's host cores
  fwd_bwd       the training-step variant (configs[2]): forward + L1 loss + backward
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
from robosimgs_amd.rendering import rasterization  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)
RASTER_PMC_TRAFFIC_BYTES = 315_551_744   # 2 x FETCH_SIZE + WRITE_SIZE of raster_fwd_kernel<3,false>, config 2 (profiles/r1/11)


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    # N > 1: what rank 0 collects.  "u8" (default) = the 8-bit images a dataset writer stores,
    # quantised on the device inside the timed region ("or gather uint8 RGB"); "fp32" =
    # the raw renders, 4x the bytes (7 x 24.9 MB per step into rank 0 at 8 ranks: about one frame time
    # of xGMI bandwidth).  The render itself is fp32 either way.
    ap.add_argument("--gather-dtype", choices=("fp32", "u8"), default="u8")
    ap.add_argument("--gather-batch", type=int, default=4, help="frames per collective (N > 1)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent frames in flight, one HIP stream + one HIP graph each")
    # debugging aid for the N > 1 control flow on a single-GPU box: all ranks share cuda:0 and the
    # collectives run over gloo with host staging.  Never used for a reported number.
    ap.add_argument("--debug-single-device-gloo", action="store_true")
    # debugging aid: run the gather code path in a world of one (checks the RCCL plumbing on a 1-GPU box)
    ap.add_argument("--force-gather", action="store_true")
    return ap.parse_args()


def barrier_sync(use_dist):
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world == 1 and a.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    debug_gloo = a.debug_single_device_gloo
    if debug_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_gather
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if debug_gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H, deg = a.width, a.height, a.sh_degree
    scene = synthetic_scene(a.n, a.log_scale_mean, deg, seed=0)
    theta = 0.3 + 2.0 * math.pi * rank / world
    cam = camera_ring(1, W, H, thetas=[theta])[0]
    t = scene.to_torch(dev, deg)
    vm = torch.from_numpy(cam.viewmat().astype(np.float32)).to(dev)[None]
    K = torch.from_numpy(cam.K.astype(np.float32)).to(dev)[None]
    tile_w, tile_h = -(-W // 16), -(-H // 16)

    def forward(cap=None, bounds="tight"):
        return rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"],
                             vm, K, W, H, sh_degree=deg, render_mode="RGB", isect_capacity=cap,
                             tile_bounds=bounds)

    # sizing passes (read n_isect back, outside every timed region).  n_isect is the classic
    # mean +- radius count SURVEY.md 8(d) calibrates (5,019,7xx at config 2) and the algorithmic
    # bytes are priced on; the frames run on the tightened lists (n_isect_binned, same image).
    colors, alphas, meta = forward(bounds="classic")
    torch.cuda.synchronize()
    n_isect = int(meta["n_isects"][0])
    n_vis = int((meta["radii"] > 0).sum())
    colors_t, alphas_t, meta_t = forward()
    assert torch.equal(colors_t, colors) and torch.equal(alphas_t, alphas), "tight tile bounds changed the image"
    n_isect_binned = int(meta_t["n_isects"][0])
    del colors_t, alphas_t, meta_t
    cap = int(n_isect_binned * 1.25) + 4096

    # ---- forward frames through the library's FrameRenderer ---------------------------------
    # One HIP graph per in-flight slot, camera in device buffers, `inflight` independent frames
    # on their own streams: the latency-bound binning kernels of one frame run under the
    # VALU-bound raster of another.  Every step submits one whole frame (projection + binning +
    # raster) and the timed region ends with a full sync.
    from robosimgs_amd import FrameRenderer, frame_to_u8
    n_fl = max(1, a.inflight)
    fr = FrameRenderer(t, W, H, render_mode="RGB", frames_in_flight=n_fl, isect_capacity=cap)
    vm_np, K_np = vm[0].cpu().numpy(), K[0].cpu().numpy()
    vm_dev, K_dev = vm[0].contiguous(), K[0].contiguous()
    cam_dev = FrameRenderer.pack_camera(vm_dev, K_dev)      # one device tensor: one copy per submit

    do_gather = use_dist and not a.no_gather
    comm_dev = "cpu" if debug_gloo else dev
    g_u8 = a.gather_dtype == "u8"
    g_dtype = torch.uint8 if g_u8 else torch.float32
    # Frames leave in batches of `gather_batch` through a double-buffered staging area: the frame
    # is converted (u8) or copied (fp32) into its place in the batch, the slot is released at once,
    # and every gather_batch-th frame one collective ships the whole batch -- a per-frame
    # collective costs ~80 us of launch / stream hand-over each, a quarter of a frame time.
    GB = max(1, a.gather_batch)
    batch_shape = (GB, H, W, 3)
    staging = [torch.empty(batch_shape, device=dev, dtype=g_dtype) for _ in range(2)] if do_gather else None
    host_staging = ([torch.empty(batch_shape, device="cpu", dtype=g_dtype) for _ in range(2)]
                    if do_gather and debug_gloo else None)          # gloo debugging mode only
    gather_bufs = None
    if do_gather and rank == 0:
        gather_bufs = [[torch.empty(batch_shape, device=comm_dev, dtype=g_dtype) for _ in range(world)]
                       for _ in range(2)]
    pending = [None, None]
    state = {"cur": 0, "fill": 0}
    tickets = []

    def ship():
        """One collective for the frames staged so far (stream-ordered after their conversion)."""
        cur = state["cur"]
        src = staging[cur]
        if debug_gloo:
            host_staging[cur].copy_(src)                           # synchronous host copy
            src = host_staging[cur]
        pending[cur] = dist.gather(src, gather_bufs[cur] if rank == 0 else None, dst=0, async_op=True)
        state["cur"], state["fill"] = cur ^ 1, 0

    def retire():
        """Fetch the oldest frame; with N > 1 stage it for the (asynchronous) RCCL gather."""
        tk = tickets.pop(0)
        f = fr.fetch(tk, check=False)
        if do_gather:
            cur, j = state["cur"], state["fill"]
            if j == 0 and pending[cur] is not None:    # the collective that last read this staging buffer
                pending[cur].wait()                    # (NCCL: the current STREAM waits, not the host)
                pending[cur] = None
            if g_u8:                                   # quantise on the device, inside the timed region
                frame_to_u8(f["colors"], f["alphas"], out=staging[cur][j].view(-1, 3))
            else:
                staging[cur][j].copy_(f["colors"], non_blocking=True)
            state["fill"] = j + 1
            if state["fill"] == GB:
                ship()
        fr.release(tk)

    def step(i):
        if len(tickets) == n_fl:
            retire()
        tickets.append(fr.submit(cam_dev))

    def drain():
        while tickets:
            retire()
        if do_gather and state["fill"] > 0:            # a partial last batch still travels (whole buffer)
            ship()
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for i in range(a.warmup):
        step(i)
    drain()
    barrier_sync(use_dist)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    drain()
    barrier_sync(use_dist)
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], device=comm_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # single-frame latency (one slot, nothing else in flight), for reference
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(20):
        tk = fr.submit(vm_dev, K_dev)
        fr.fetch(tk, check=False)
        fr.release(tk)
        torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t1) / 20 * 1e3
    outs = [(None, None, s["meta"]) for s in fr._slots]
    if do_gather and rank == 0:
        # the collective really delivered every rank's frame (all cameras see the scene)
        for r_ in range(world):
            assert float(gather_bufs[0][r_][0].float().abs().max()) > 0.0, f"rank {r_}: empty gathered frame"
    status = max(int(o[2]["isect_status"].max().item()) for o in outs)
    assert status == 0, "tile-intersection capacity overflow inside the timed region"
    frames_per_s = world * a.steps / elapsed
    ms_per_step = elapsed / a.steps * 1e3

    result = {
        "metric": "frames/sec + ms/frame (fwd, fwd+bwd) at 1M Gaussians 1920x1080",
        "value": round(frames_per_s, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[1]: {a.n} Gaussians, SH degree {deg}, {W}x{H} forward "
                               "render, one camera per GPU",
                   "n_gaussians": a.n, "n_visible": n_vis, "n_isect": n_isect,
                   "n_isect_binned": n_isect_binned,
                   "tiles": tile_w * tile_h, "cameras_per_step": world,
                   "gather": (("8-bit RGB images (frame_to_u8 on the device, inside the timed region)" if g_u8
                               else "fp32 RGB frames") + f" to rank 0 (RCCL), {GB} frames per collective") if do_gather else "none",
                   "launch": f"one HIP graph per frame, no host read-back, {n_fl} independent "
                             "frames in flight on separate HIP streams",
                   "frames_in_flight": n_fl, "single_frame_latency_ms": round(latency_ms, 4)},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (tile raster forward), HIP events on the stream
        radii, m2d, depths, con, _, feats, splats = ops.project_color_fwd_raw(
            t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"], vm[0], K[0], W,
            H, 0.3, 0.01, 1e10, 0.0, False, False, want_splats=True)
        tl = ops.isect_tiles_raw(m2d, radii, depths, tile_w, tile_h, cap, want_tiles_per_gauss=False,
                                 conics=con, opacities=t["opacities"])
        # the inference variant (no last_ids), i.e. the kernel the timed frames above run
        out = None
        reps = 50
        for _ in range(5):
            out = ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tile_w, tile_h,
                                        tl.tile_offsets, tl.flatten_ids, out=out, track_last=False,
                                        splats=splats)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.rasterize_fwd_raw(m2d, con, feats, t["opacities"], None, W, H, tile_w, tile_h,
                                  tl.tile_offsets, tl.flatten_ids, out=out, track_last=False,
                                  splats=splats)
        e1.record()
        torch.cuda.synchronize()
        raster_ms = e0.elapsed_time(e1) / reps
        n_px = W * H
        algo_bytes = n_isect * 44 + n_px * 24 + tile_w * tile_h * 8     # SURVEY.md 8(d) (kept as is: the
        # inference variant skips the 4 B/px last_ids store the formula includes)
        achieved = algo_bytes / (raster_ms * 1e-3) / 1e9
        result["roofline"] = {"kernel": "raster_fwd_kernel<3, false>", "bound": "hbm",
                              "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 4),
                              # PMC passes cannot run inside bench.py; value measured with
                              # scripts/pmc.sh on this kernel and workload (profiles/r1/11)
                              "traffic": RASTER_PMC_TRAFFIC_BYTES if (a.n, W, H, deg) == (1_000_000, 1920, 1080, 3) else None,
                              "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes: "
                                                "2 x 140.3 MB (gfx950 counts 128-byte requests at 64 B; calibrated "
                                                "on this gather pattern) + 34.9 MB per launch (profiles/r1/11_pmc_final.md)",
                              "algorithmic_bytes": algo_bytes,
                              "kernel_ms": round(raster_ms, 4),
                              "valu_busy_frac": 0.92,
                              "note": "VALU-bound kernel: SQ_ACTIVE_INST_VALU = 92 % of SIMD cycles "
                                      "(profiles/r1/11); the HBM fraction is reported as the "
                                      "contract asks; see DESIGN.md 4.3"}

        # the second-largest forward kernel is HBM-bound: projection + SH colour (SURVEY.md 8(d):
        # N*236 + n_vis*48 algorithmic bytes), timed the same way
        for _ in range(5):
            ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"],
                                      vm[0], K[0], W, H, 0.3, 0.01, 1e10, 0.0, False, False, want_splats=True)
        e0.record()
        for _ in range(reps):
            ops.project_color_fwd_raw(t["means"], t["quats"], t["scales"], t["opacities"], deg, t["colors"],
                                      vm[0], K[0], W, H, 0.3, 0.01, 1e10, 0.0, False, False, want_splats=True)
        e1.record()
        torch.cuda.synchronize()
        proj_ms = e0.elapsed_time(e1) / reps
        proj_bytes = a.n * (44 + 12 * (deg + 1) ** 2) + n_vis * 48
        result["roofline_projection"] = {
            "kernel": f"project_color_fwd_kernel<{deg}, {'true' if deg >= 2 else 'false'}>", "bound": "hbm",
            "achieved": round(proj_bytes / (proj_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(proj_bytes / (proj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes": proj_bytes, "kernel_ms": round(proj_ms, 4),
            "note": "also writes the 48-byte splat records (n_vis * 48 more bytes, not counted)"}

        # ---- training-step variant (configs[2]): forward + L1 + backward ------------------
        try:
            result["fwd_bwd"] = bench_fwd_bwd(a, t, vm, K, W, H, deg, cap, dev)
        except Exception as e:  # keep the headline line even if this leg fails
            result["fwd_bwd"] = {"error": repr(e)[:200]}

        # ---- CPU baseline: the oracle's C++/OpenMP port on this box's host cores -----------
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene, cam, W, H, deg, a.cpu_seconds)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_fwd_bwd(a, t, vm, K, W, H, deg, cap, dev):
    from robosimgs_amd import l1_loss
    names = ("means", "quats", "scales", "opacities", "colors")
    params = {k: t[k].detach().clone().requires_grad_(True) for k in names}
    target = torch.rand(1, H, W, 3, device=dev, generator=torch.Generator(dev).manual_seed(1))

    def train_step():
        for p in params.values():
            p.grad = None
        colors, alphas, meta = rasterization(params["means"], params["quats"], params["scales"],
                                             params["opacities"], params["colors"], vm, K, W, H,
                                             sh_degree=deg, render_mode="RGB", isect_capacity=cap)
        loss = l1_loss(colors, target)      # fused HIP L1 (== (colors - target).abs().mean())
        loss.backward()
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
    t0 = time.perf_counter()
    for _ in range(a.bwd_steps):
        runner()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.bwd_steps
    return {"workload": "configs[2]: forward + L1 loss to U(0,1) target (seed 1) + backward",
            "ms_per_step": round(dt * 1e3, 4), "steps_per_s": round(1.0 / dt, 2),
            "steps": a.bwd_steps, "launch": mode}


def cpu_baseline(scene, cam, W, H, deg, budget_s):
    from oracle import cpu_ref
    threads = cpu_ref.max_threads()
    args = (scene.means, scene.quats, scene.scales, scene.opacities, scene.sh_coeffs,
            cam.viewmat(), cam.K, W, H, deg)
    cpu_ref.render(*args)                                   # warm-up (page-in, thread pool)
    times = []
    t_start = time.perf_counter()
    while (time.perf_counter() - t_start) < budget_s and len(times) < 50:
        t0 = time.perf_counter()
        _, _, info = cpu_ref.render(*args)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} full frames of the same 1M-Gaussian 1080p workload after 1 "
                      f"warm-up, median {med * 1e3:.1f} ms/frame; oracle/gs_cpu.cpp, OpenMP, "
                      f"{threads} threads on {model}",
            "pair_evals": info["pair_evals"]}


if __name__ == "__main__":
    main()
