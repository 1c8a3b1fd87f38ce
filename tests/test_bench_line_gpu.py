"""-m gpu: the contract of bench.py's stdout -- the LAST line is one compact JSON object the driver can parse out of an 8-KB
tail (round 4 printed 21 KB and went unmeasured), carrying the headline, the dominant-kernel roofline and the CPU baseline;
everything else is in the detail file."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_last_stdout_line_is_small_and_complete(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    detail = str(tmp_path / "detail.json")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8", "--legs", "ref_split",
                        "--detail", detail], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 3072, len(last)
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "stages"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 0 and line["finite_output"] is True
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and 0.0 < roof["frac"] < 1.0 and roof["launches"] > 0 and roof["avg_launch_ms"] > 0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-2)
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "1 pair" in cb["sample"] and "median of 3" in cb["sample"]
    assert set(cb["stage_seconds"]) == {"encoder", "heads", "matcher", "solver", "total"}
    assert line["sustained_60"] > 0
    assert 1 <= len(line["stages"]) <= 6
    assert line["value_ref_precision"] > 0 and line["single_pair_ms"] > 0
    full = json.load(open(detail))
    assert "legs" in full and "ref_split" in full["legs"] and len(full["roofline"]["stages"]) >= 6
    assert full["sustained"]["steps"] == 60 and len(full["cpu_baseline"]["runs_total_seconds"]) == 3
    assert full["cpu_baseline"]["threads_probe"]["measured_once"]["seconds_threads_256"] > 100   # (the all-cores probe is opt-in)
    assert "split" in full["legs"]["ref_split"]["heads_operands"]
