"""-m gpu: the multi-GPU path on the hardware that is there -- ONE GPU, world size 1, backend nccl (= RCCL): process-group
init, the pose all-gather, the side-stream gatherer, the sharded evaluation feed, and bench.py launched the way the driver
launches it for N > 1 (torch.distributed.run, rendezvous on 127.0.0.1).  The first 8-GPU run is then not the first time
this code meets RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script_args, timeout=900):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port())] + script_args
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def test_rccl_world1_gather_paths():
    out = _torchrun([os.path.join(ROOT, "tests", "helpers", "rccl_world1.py")])
    line = [ln for ln in out.splitlines() if ln.startswith("RCCL_WORLD1 ")][-1]
    res = json.loads(line[len("RCCL_WORLD1 "):])
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["sharded_equal"] and res["gatherer_equal"] and res["side_stream_is_not_current"] and res["junk_finite"]
    assert res["empty_slice_shapes"] == [[0, 3, 3], [0, 1, 3], [0, 1]]
    assert res["eval_zip_equal"]


def test_bench_under_torchrun_world1(tmp_path):
    """bench.py as the driver launches it for N > 1, with N = 1: one rank, RCCL initialised, the per-step all-gather on the
    side stream, MAX-reduced time; the gathered poses of the last step equal an un-distributed forward bit for bit."""
    dump = str(tmp_path / "poses.pt")
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--lean",
                     "--graph", "off", "--dump-poses", dump])
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{\"metric\"")][-1])
    assert line["n_gpus"] == 1 and line["finite_output"] and line["steps"] == 2 and line["value"] > 0
    assert "all-gather" in line["config"]["parallelism"]
    got = torch.load(dump)
    from mickey_amd import synthetic as syn
    from mickey_amd.config import default_cfg
    from mickey_amd.model import MickeyRelativePose
    dev = torch.device("cuda:0")
    cfg = default_cfg()
    cfg["AMD"]["ENCODER_DTYPE"] = "bf16"
    cfg["AMD"]["SEED"] = 0
    cfg["AMD"]["GRAPH"] = False
    m = MickeyRelativePose(cfg)
    m.load_state_dict(syn.mickey_state_dict(cfg, seed=0))
    m = m.to(dev)
    data0 = {k: v.to(dev) for k, v in syn.synthetic_batch(B=2, H=720, W=540, seed=1234).items()}
    for _ in range(3):   # 1 warm-up + 2 timed steps: the Philox stream offset advances with every forward
        d = dict(data0)
        R, t = m(d)
    assert torch.equal(got["R"].cpu(), R.cpu()) and torch.equal(got["t"].cpu(), t.cpu())
