"""Camera-sharded multi-GPU rendering: one process per GPU, the scene
replicated on every rank (236 MB at 1 M Gaussians), cameras split in contiguous blocks, no
communication during a render.  The only collectives are the ones the path really has:
  * gather of finished frames onto one rank (novel-view generation), and
  * all-reduce of parameter gradients (training step).
`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" with CPU
tensors, which is how the sharding / gather logic is tested without GPUs.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_sizes(n_cameras: int, world_size: int, weights: Optional[Sequence[float]] = None) -> List[int]:
    """Cameras per rank.  weights (one per rank, > 0; None = equal): shares in proportion, rounded by largest remainder
    (ties to the lower rank), so the sizes always add up to n_cameras.  The rank that also RECEIVES the gathered frames
    (and converts its own) gets a weight below 1: at 8 ranks it takes 7/8 of every frame over its seven xGMI links."""
    if weights is None:
        base, rem = divmod(n_cameras, world_size)
        return [base + (1 if r < rem else 0) for r in range(world_size)]
    w = [float(x) for x in weights]
    if len(w) != world_size or min(w) <= 0:
        raise ValueError(f"need {world_size} positive weights, got {weights}")
    exact = [n_cameras * x / sum(w) for x in w]
    sizes = [int(e) for e in exact]
    order = sorted(range(world_size), key=lambda r: (-(exact[r] - sizes[r]), r))
    for r in order[:n_cameras - sum(sizes)]:
        sizes[r] += 1
    return sizes


def shard_cameras(n_cameras: int, world_size: int, rank: int, weights: Optional[Sequence[float]] = None) -> range:
    """Contiguous block of camera indices owned by `rank`; equal blocks differ by at most one, weighted blocks
    (shard_sizes) follow the weights."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    sizes = shard_sizes(n_cameras, world_size, weights)
    start = sum(sizes[:rank])
    return range(start, start + sizes[rank])


def root_weights(world_size: int, root_weight: float = 1.0, dst: int = 0) -> Optional[List[float]]:
    """Weights for shard_cameras with the gathering rank at `root_weight` and every other rank at 1 (None when equal)."""
    if world_size < 2 or root_weight == 1.0:
        return None
    return [root_weight if r == dst else 1.0 for r in range(world_size)]


def gather_frames(local: torch.Tensor, n_cameras: int, dst: int = 0,
                  group=None, weights: Optional[Sequence[float]] = None) -> Optional[torch.Tensor]:
    """Gather per-rank frame blocks [c_r, H, W, D] onto `dst` in camera order.

    Ragged shards (n_cameras % world != 0) are padded to the largest shard for the collective
    and trimmed afterwards.  Returns [n_cameras, H, W, D] on `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_cameras, world, weights)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} frames, shard is {sizes[rank]}")
    biggest = max(sizes)
    send = local.contiguous()
    if send.shape[0] < biggest:
        pad = torch.zeros((biggest - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype,
                          device=send.device)
        send = torch.cat([send, pad], dim=0)
    bufs = None
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def all_reduce_gradients(params: Sequence[torch.Tensor], group=None, average: bool = True) -> None:
    """Sum (or average) .grad of every parameter over ranks, one flat bucket per call: 59
    floats per Gaussian = 236 MB at 1 M, a size at which the collective is bandwidth-bound."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def render_sharded(tensors: dict, viewmats: torch.Tensor, Ks: torch.Tensor, width: int,
                   height: int, dst: int = 0, group=None, gather: bool = True, renderer=None,
                   as_u8: bool = False, u8_background=None, weights: Optional[Sequence[float]] = None,
                   **kw) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor], range]:
    """Render this rank's block of the C cameras and (optionally) gather all frames on `dst`.
    `renderer`: a persistent `FrameRenderer` built for this scene / resolution / render mode;
    when given (inference only), the rank's cameras go through it with several frames in flight
    instead of one `rasterization` call per block.

    tensors: dict(means, quats, scales, opacities, colors, sh_degree) as from
    Gaussians.to_torch().  viewmats [C,4,4] / Ks [C,3,3] hold ALL cameras on every rank.
    Returns (colors, alphas, my_range): full [C,...] tensors on `dst` when gathered, this
    rank's block otherwise.  as_u8=True gathers 8-bit RGB images instead of fp32 renders
    (compositing.frame_to_u8 with `u8_background`; alphas are then not gathered): a quarter of the
    bytes over xGMI, which is what a dataset writer stores anyway.
    weights: camera shares per rank (shard_sizes; root_weights(world, 0.5) halves the gathering rank's share).
    A `renderer` built with dataset_output= (and the default dataset_keep_float=False) holds no float frame: the two
    returned tensors are then the dataset frames, rgba uint8 [C,H,W,4] and ray distance [C,H,W,1], gathered alike."""
    from .rendering import rasterization
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    C = viewmats.shape[0]
    mine = shard_cameras(C, world, rank, weights)
    sel = slice(mine.start, mine.stop)
    if len(mine) and renderer is not None:
        cs, als = [None] * len(mine), [None] * len(mine)
        tickets, nxt = [], 0
        for i in range(len(mine)):
            while nxt < len(mine) and len(tickets) < renderer.n_slots:
                tickets.append(renderer.submit(viewmats[mine.start + nxt], Ks[mine.start + nxt]))
                nxt += 1
            tk = tickets.pop(0)
            f = renderer.fetch(tk)
            if f["colors"] is None:           # a renderer that writes dataset frames only (dataset_output=): RGBA8 + distance
                cs[i], als[i] = f["rgba"].clone(), f["distance"].clone()
            else:
                cs[i], als[i] = f["colors"].clone(), f["alphas"].clone()
            renderer.release(tk)
        colors, alphas = torch.stack(cs), torch.stack(als)
    elif len(mine):
        colors, alphas, _ = rasterization(tensors["means"], tensors["quats"], tensors["scales"],
                                          tensors["opacities"], tensors["colors"], viewmats[sel],
                                          Ks[sel], width, height,
                                          sh_degree=tensors.get("sh_degree"), **kw)
    elif renderer is not None and renderer.dataset_dtype is not None and not renderer.dataset_keep_float:
        colors = torch.zeros(0, height, width, 4, dtype=torch.uint8, device=viewmats.device)
        alphas = torch.zeros(0, height, width, 1, dtype=renderer.dataset_dtype, device=viewmats.device)
    else:
        d = {"RGB": 3, "D": 1, "ED": 1}.get(kw.get("render_mode", "RGB"), 4)   # RGB+D / RGB+ED: 4
        colors = torch.zeros(0, height, width, d, device=viewmats.device)
        alphas = torch.zeros(0, height, width, 1, device=viewmats.device)
    if as_u8 and colors.dtype == torch.uint8:
        raise ValueError("as_u8 quantises float frames; this renderer already returns RGBA8 dataset frames")
    if as_u8:
        from .compositing import frame_to_u8
        colors = (frame_to_u8(colors, alphas, u8_background) if len(mine)
                  else torch.zeros(0, height, width, 3, dtype=torch.uint8, device=viewmats.device))
    if not gather or world == 1:
        return colors, alphas, mine
    return (gather_frames(colors, C, dst, group, weights),
            None if as_u8 else gather_frames(alphas, C, dst, group, weights), mine)
