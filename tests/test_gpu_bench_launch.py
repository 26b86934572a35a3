"""bench.py --gpus N started the way the driver starts the N = 1 run -- plain `python bench.py ...`, no
torch.distributed.run, no WORLD_SIZE -- must launch its own ranks and print exactly one JSON line.
Two ranks share cuda:0 over gloo here (the control flow of configs[3]: ring sharding, batched gather, MAX
over ranks); the RCCL run on N GPUs is the driver's."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_python_bench_gpus_2_launches_its_ranks_and_prints_one_json_line():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-single-device-gloo",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--min-seconds", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["world_size"] == 2
    assert res["config"]["frames_per_rank"] == [32, 32]
    assert res["scaling"] == "strong" and res["steps"] == 2 and res["warmup"] == 1
    assert res["config"]["frames_per_step_all_ranks"] == 64
    assert res["value"] > 0 and abs(res["value"] - 64 * 2 / (res["ms_per_step"] * 2 * 1e-3)) / res["value"] < 1e-3


def test_gpus_n_without_enough_devices_fails_with_a_clear_message():
    """No GPU needed: on this box (0 or 1 device) --gpus 8 must say what is missing, not die in a rendezvous."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU node: the launch would succeed")
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs 8 visible GPUs" in r.stderr
