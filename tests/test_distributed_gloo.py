"""N > 1 path on CPU: two gloo ranks exercise the camera sharding, the (ragged) frame gather
and the gradient all-reduce that bench.py / render_sharded use with RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from robosimgs_amd.distributed import (all_reduce_gradients, gather_frames, root_weights, shard_cameras,
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


def test_weighted_shards_follow_the_weights_and_partition_exactly():
    """The gathering rank renders a smaller block (it also receives (N - 1) / N of every frame): shares in proportion
    to the weights, largest remainder, contiguous, always a partition."""
    assert shard_sizes(64, 8, root_weights(8, 0.5)) == [4, 9, 9, 9, 9, 8, 8, 8]
    assert shard_sizes(64, 2, root_weights(2, 0.5)) == [21, 43]
    assert shard_sizes(64, 4, root_weights(4, 0.5)) == [9, 19, 18, 18]
    assert root_weights(8, 1.0) is None and root_weights(1, 0.5) is None
    assert shard_sizes(64, 8, root_weights(8, 1.0)) == [8] * 8
    for n in (0, 1, 5, 64, 65):
        for world in (2, 3, 8):
            for w0 in (0.25, 0.5, 0.8, 2.0):
                w = root_weights(world, w0)
                blocks = [shard_cameras(n, world, r, w) for r in range(world)]
                assert [i for b in blocks for i in b] == list(range(n))
                sizes = shard_sizes(n, world, w)
                exact = [n * x / sum(w) for x in w]
                assert all(abs(s_ - e) < 1 for s_, e in zip(sizes, exact))
    with pytest.raises(ValueError):
        shard_sizes(8, 2, [1.0, 0.0])
    with pytest.raises(ValueError):
        shard_sizes(8, 2, [1.0])


def _worker(rank, world, port, n_cams, out, root_weight=1.0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = root_weights(world, root_weight)
        mine = shard_cameras(n_cams, world, rank, w)
        frames = torch.stack([torch.full((4, 6, 3), float(c)) for c in mine]) if len(mine) else \
            torch.zeros(0, 4, 6, 3)
        got = gather_frames(frames, n_cams, dst=0, weights=w)
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


@pytest.mark.parametrize("n_cams,world,root_weight", [(5, 2, 1.0), (8, 2, 1.0), (8, 2, 0.5), (10, 3, 0.5)])
def test_gather_and_allreduce_over_gloo(n_cams, world, root_weight):
    """Worlds of 2 and 3, equal and weighted (ragged) shards: the gather returns every camera once, in order."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_cams, q, root_weight)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == {r: "ok" for r in range(world)}, res
