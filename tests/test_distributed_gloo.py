"""N > 1 path on CPU: two gloo ranks exercise the camera sharding, the (ragged) frame gather
and the gradient all-reduce that bench.py / render_sharded use with RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from robosimgs_amd.distributed import (all_reduce_gradients, gather_frames, shard_cameras,
                                       shard_sizes)


def test_shard_cameras_partitions_exactly():
    for n in (0, 1, 7, 8, 63, 64, 65):
        for world in (1, 2, 3, 8):
            blocks = [shard_cameras(n, world, r) for r in range(world)]
            assert [i for b in blocks for i in b] == list(range(n))
            sizes = shard_sizes(n, world)
            assert max(sizes) - min(sizes) <= 1
    assert list(shard_cameras(64, 8, 3)) == list(range(24, 32))      # BASELINE configs[3]: 8 views/GPU
    with pytest.raises(ValueError):
        shard_cameras(4, 2, 2)


def _worker(rank, world, port, n_cams, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_cameras(n_cams, world, rank)
        frames = torch.stack([torch.full((4, 6, 3), float(c)) for c in mine]) if len(mine) else \
            torch.zeros(0, 4, 6, 3)
        got = gather_frames(frames, n_cams, dst=0)
        if rank == 0:
            assert got.shape == (n_cams, 4, 6, 3)
            assert torch.equal(got[:, 0, 0, 0], torch.arange(n_cams, dtype=torch.float32))
        else:
            assert got is None
        p = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
        p[0].grad = torch.full((5, 3), float(rank + 1))
        p[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
        all_reduce_gradients(p, average=False)
        tot = sum(range(1, world + 1))
        assert torch.equal(p[0].grad, torch.full((5, 3), float(tot)))
        assert torch.equal(p[1].grad, torch.arange(7, dtype=torch.float32) * tot)
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_cams", [5, 8])
def test_two_rank_gather_and_allreduce(n_cams):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_cams, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == {0: "ok", 1: "ok"}, res
