"""bench.py's multi-GPU plumbing without a GPU: --gpus N must launch N ranks itself (torch.distributed.run on
127.0.0.1), report the world size the process group saw, and refuse loudly what it cannot honour."""
import argparse
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(gpus):
    return argparse.Namespace(gpus=gpus)


def test_resolve_world_rules():
    assert bench.resolve_world(_args(1), {}, 1) == ("single", 1)
    assert bench.resolve_world(_args(2), {}, 8) == ("spawn", 2)
    assert bench.resolve_world(_args(8), {"RANK": "3", "WORLD_SIZE": "8"}, 8) == ("rank", 8)
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(_args(2), {}, 1)          # python bench.py --gpus 2 on a 1-GPU box
    assert "only 1 GPU" in str(e.value)
    with pytest.raises(SystemExit):
        bench.resolve_world(_args(1), {}, 0)
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(_args(4), {"RANK": "0", "WORLD_SIZE": "2"}, 8)   # torchrun of 2 ranks but --gpus 4
    assert "must agree" in str(e.value)
    with pytest.raises(SystemExit):
        bench.resolve_world(_args(8), {"RANK": "0", "WORLD_SIZE": "8"}, 4)


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--backend", "gloo", "--batch", "3", "--steps", "2",
                        "--warmup", "1"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, timeout=300)
    return r


def test_gpus_2_spawns_two_ranks_under_gloo():
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["stub"] is True and out["config"]["global_batch"] == 6 and out["steps"] == 2


def test_gpus_8_stub_reports_every_rank():
    """The world size the driver's scaling run ends at: 8 ranks launch, one line, per-rank step times (min / max / all)
    and each rank's CPU placement (an even split of the allowed cores here: no NUMA information without a GPU)."""
    r = _run(["--gpus", "8"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 24
    pr = out["per_rank_ms_per_step"]
    assert len(pr["all"]) == 8 and pr["min"] <= pr["max"] and abs(pr["max"] - out["ms_per_step"]) < 1e-6
    aff = out["config"]["affinity"]
    assert len(aff) == 8 and all("n_cpus" in a and a["n_cpus"] >= 1 for a in aff)
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:
        assert sum(a["n_cpus"] for a in aff) == ncpu      # a partition of the allowed cores


def test_gpus_1_is_single_process():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1


def test_launcher_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "4"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_real_bench_refuses_more_gpus_than_visible():
    """Without --stub: this container has no GPU, so --gpus 2 must exit non-zero with a clear message (and must not
    silently run one process, which is what round 1 did)."""
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("2 GPUs visible here")
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


def test_compact_line_from_the_round4_detail_is_small_and_complete():
    """Round 4's line was 21 KB and the driver (8-KB tail) could not parse it.  The line is now a pure function of the detail
    dict: fed with that very 21-KB object (profiles/r04_bench_b32.json) it must come out < 3 KB, parse, and carry the
    contract's keys, the dominant-kernel roofline, the CPU baseline and a <= 6-entry stage summary."""
    detail = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_b32.json")))
    assert len(json.dumps(detail)) > 8192                      # the input really is the oversized one
    txt = bench.compact_line(detail, "gpurun_out/bench_detail.json")
    assert "\n" not in txt and len(txt) < bench.LINE_LIMIT <= 3072
    line = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == pytest.approx(detail["value"], rel=1e-3) and line["ms_per_step"] == pytest.approx(detail["ms_per_step"], rel=1e-3)
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
              "launches", "traffic_source"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], rel=1e-3)
    for k in ("value", "unit", "cores", "cores_available", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert 1 <= len(line["stages"]) <= 6 and all(set(s) == {"stage", "ms_per_step", "frac"} for s in line["stages"])
    assert line["stages"][0]["stage"] == "encoder_gemm"        # sorted by time
    assert line["config"]["workload"] and "affinity" not in line["config"]
    assert line["value_ref_precision"] == pytest.approx(detail["legs"]["ref_split"]["value"], rel=1e-3)
    assert line["single_pair_ms"] == pytest.approx(detail["single_pair"]["ms_per_pair"], rel=1e-3)
    assert "legs" not in line and "precision" not in line and "alt" not in line and "precision_matched" not in line


def test_stub_line_is_last_and_small():
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 3072 and json.loads(last)["n_gpus"] == 2


def test_default_arguments_keep_the_driver_run_short():
    """The default run = headline + 60 sustained steps + ref_split and natural-operands legs + one-pair leg + CPU baseline (59 s on
    the box, profiles/r06i_*); everything else is opt-in."""
    a = bench.parse_args([])
    assert a.leg_set == {"ref_split", "natural"} and a.sustained and not a.include_h2d and not a.precision
    assert not bench.parse_args(["--no-sustained"]).sustained
    assert bench.parse_args(["--legs", "all"]).leg_set >= {"fp16", "ref_split", "vit_small", "config5"}
    lean = bench.parse_args(["--lean"])
    assert lean.leg_set == set() and lean.no_cpu_baseline and lean.no_single and not lean.sustained
    with pytest.raises(SystemExit):
        bench.parse_args(["--legs", "nonsense"])


def test_compact_line_at_eight_gpus_stays_small():
    """The N = 8 line carries min / max of the ranks' own step times and drops the per-rank placements: still < 3 KB."""
    detail = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_b32.json")))
    detail["n_gpus"] = 8
    detail["per_rank_ms_per_step"] = {"min": 115.1, "max": 117.2, "all": [116.0] * 8}
    detail["config"]["affinity"] = [{"numa_node": r // 4, "cpus": list(range(32 * r, 32 * r + 32)), "n_cpus": 32} for r in range(8)]
    txt = bench.compact_line(detail, "gpurun_out/bench_detail.json")
    assert len(txt) < bench.LINE_LIMIT
    line = json.loads(txt)
    assert line["n_gpus"] == 8 and line["per_rank_ms_per_step"] == {"min": 115.1, "max": 117.2} and "affinity" not in line["config"]
