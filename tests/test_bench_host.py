"""Host-side logic of bench.py that needs no GPU: the PMC traffic figure is printed only when it was
measured on THIS build of libmgs.so, and the N > 1 camera sharding of configs[3]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_pmc_traffic_is_keyed_to_the_build_stamp(tmp_path, monkeypatch):
    import bench
    from robosimgs_amd.csrc import build as hip_build
    stamp = hip_build.current_stamp()
    rec = {"stamp": stamp, "kernels": {"raster_fwd": {"traffic_bytes": 123456789, "source": "test"}}}
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "PMC_FILE", str(f))
    assert bench.pmc_traffic("raster_fwd", True) == (123456789, "test")
    assert bench.pmc_traffic("raster_fwd", False)[0] is None            # another workload: no figure applies
    assert bench.pmc_traffic("no_such_kernel", True)[0] is None
    rec["stamp"] = "0" * 64                                              # measured on another build: stale
    f.write_text(json.dumps(rec))
    val, why = bench.pmc_traffic("raster_fwd", True)
    assert val is None and "stale" in why
    monkeypatch.setattr(bench, "PMC_FILE", str(tmp_path / "missing.json"))
    assert bench.pmc_traffic("raster_fwd", True)[0] is None


def test_committed_pmc_record_has_the_kernels_bench_asks_for():
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert len(rec["stamp"]) == 64
    for k in ("raster_fwd", "raster_fwd_q"):
        assert rec["kernels"][k]["traffic_bytes"] > 100_000_000


def test_ring_sharding_of_configs3():
    from robosimgs_amd.distributed import shard_cameras
    for world in (1, 2, 3, 4, 8):
        blocks = [list(shard_cameras(64, world, r)) for r in range(world)]
        assert sum(blocks, []) == list(range(64))
        assert max(map(len, blocks)) - min(map(len, blocks)) <= 1
    assert list(shard_cameras(64, 8, 3)) == list(range(24, 32))


def test_scaling_prediction_is_min_of_render_and_root_bounds():
    """bench.py's `config.scaling_prediction`: per payload and N in (2, 4, 8), min(N x the one-rank rate, the single root's
    ceiling); fp32 frames are root-bound at 8 ranks (configs[3] as written cannot reach 6 x), the 6-byte dataset payload passes 6 x."""
    import bench
    W, H = 1920, 1080
    p = bench.scaling_prediction({"fp32": 4400.0, "dataset": 4350.0}, W, H)
    assert set(p) == set(bench.PAYLOAD_BYTES_PER_PX)
    for m, rows in p.items():
        r1 = rows["one_rank_frames_per_s"]
        assert rows["one_rank_rate_measured_with_this_payload"] == (m in ("fp32", "dataset"))
        for n in (2, 4, 8):
            row = rows["by_world_size"][str(n)]
            rb = bench.root_bound_frames_per_s(m, W, H, n)
            assert abs(row["root_bound"] - rb) <= 1 and abs(row["render_bound"] - n * r1) <= 1
            assert abs(row["predicted_frames_per_s"] - min(n * r1, rb)) <= 1
            assert row["limited_by"] == ("root" if rb < n * r1 else "render")
    assert p["fp32"]["by_world_size"]["8"]["limited_by"] == "root" and p["fp32"]["by_world_size"]["8"]["speedup_vs_one_rank_fp32"] < 6
    assert p["dataset16"]["by_world_size"]["8"]["speedup_vs_one_rank_fp32"] > 6 > p["dataset"]["by_world_size"]["8"]["speedup_vs_one_rank_fp32"]
    assert p["u8"]["by_world_size"]["8"]["limited_by"] == "render" and p["u8"]["by_world_size"]["8"]["speedup_vs_one_rank_fp32"] == 8.0
