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
    env["BENCH_WATCHDOG_S"] = "420"       # a stuck run dumps every process's stacks and ends before the timeout below
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-single-device-gloo",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-stress", "--min-seconds", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["world_size"] == 2
    # configs[3] as BASELINE.json writes it: EQUAL blocks ("sharded 8 views/GPU": 32 + 32 at two ranks) and `value` on the
    # fp32 RGB + expected depth + alpha gather, the payload that carries 1e-4 to the root
    cfg = res["config"]
    assert cfg["frames_per_rank"] == [32, 32] and cfg["root_weight"] == 1.0
    assert "rccl_version" in cfg and cfg["rccl_version"] is None       # gloo here; over nccl bench.py itself asserts a version string
    assert sum(cfg["frames_per_rank"]) == 64 and all(sum(v["frames_per_rank"]) == 64 for v in cfg["alternates"].values())
    sp = cfg["scaling_prediction"]
    assert sp["fp32"]["one_rank_rate_measured_with_this_payload"] and set(sp["fp32"]["by_world_size"]) == {"2", "4", "8"}
    assert all(r["predicted_frames_per_s"] == min(r["render_bound"], r["root_bound"]) for r in sp["fp32"]["by_world_size"].values())
    assert "fp32 RGB + expected depth + alpha" in cfg["gather"] and "32/32 views" in cfg["workload"]
    # the named alternates of the same run: the reference-pinned dataset frame, its fp16-distance variant, and that one
    # with a smaller block on the gathering rank (weight 1 - 0.07 (N - 1) = 0.93: 64 * 0.93 / 1.93 = 30.8 -> 31 + 33)
    alt = cfg["alternates"]
    assert list(alt) == ["dataset", "dataset16", "dataset16_weighted_root"]
    assert [alt[k]["bytes_per_frame"] for k in alt] == [8 * 1920 * 1080, 6 * 1920 * 1080, 6 * 1920 * 1080]
    assert alt["dataset"]["frames_per_rank"] == [32, 32] and alt["dataset16"]["frames_per_rank"] == [32, 32]
    assert alt["dataset16_weighted_root"]["frames_per_rank"] == [31, 33]
    assert abs(alt["dataset16_weighted_root"]["root_weight"] - 0.93) < 1e-9 and alt["dataset"]["root_weight"] == 1.0
    assert all(v["frames_per_s"] > 0 and v["ms_per_step"] > 0 and v["within_1e-4_at_the_root"] is False for v in alt.values())
    # where each rank's time went (HIP events), for every leg of the run, and the single-root ceiling by payload
    for leg, frames in (("fp32", [32, 32]), ("dataset", [32, 32]), ("dataset16", [32, 32]), ("dataset16_weighted_root", [31, 33])):
        rows = cfg["per_rank"][leg]
        assert len(rows) == 2 and [r_["frames"] for r_ in rows] == [f * 2 for f in frames], leg
        for r_ in rows:
            assert set(r_) == {"frames", "render_span_ms", "convert_ms", "after_last_convert_ms", "region_ms"}
            assert 0 < r_["render_span_ms"] <= r_["region_ms"] and r_["convert_ms"] > 0 and r_["after_last_convert_ms"] >= 0
    rb = cfg["root_bound"]
    assert rb["dataset16"]["bytes_per_frame"] == 6 * 1920 * 1080 and rb["fp32"]["bytes_per_frame"] == 20 * 1920 * 1080
    assert rb["u8"]["frames_per_s_ceiling"] > rb["dataset16"]["frames_per_s_ceiling"] > rb["dataset"]["frames_per_s_ceiling"] > rb["fp32"]["frames_per_s_ceiling"]
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
