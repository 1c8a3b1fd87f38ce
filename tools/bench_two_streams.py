"""Two forwards in flight (dev tool): batch i on HIP stream i % 2, each stream with its own model instance (own workspace and
Philox counter), against the same forwards issued back to back on one stream.  Question: do the tails of the big kernels
(1940 tiles on 256 CUs = 7.6 waves) and the launch-bound heads / matcher / solver kernels of one forward hide behind the other
forward's encoder?   python tools/bench_two_streams.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import synthetic as syn  # noqa: E402
from mickey_amd.config import default_cfg  # noqa: E402
from mickey_amd.model import MickeyRelativePose  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("B", "32")), 720, 540
cfg = default_cfg()
cfg["AMD"]["ENCODER_DTYPE"] = "bf16"
sd = syn.mickey_state_dict(cfg, seed=0, arch="vit_large")
models = []
for i in range(2):
    m = MickeyRelativePose(cfg)
    m.load_state_dict(sd)
    models.append(m.to(dev))
data = [{k: v.to(dev) for k, v in syn.synthetic_batch(B=B, H=H, W=W, seed=1234 + i).items()} for i in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def run(two, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for i in range(n):
        s = streams[i % 2] if two else streams[0]
        with torch.cuda.stream(s):
            R, t = models[i % 2](dict(data[i % 2]))
            outs.append(R)
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(o).all()) for o in outs[-2:])
    return (time.perf_counter() - t0) / n


for two in (False, True):
    run(two, 4)
res = {False: [], True: []}
for rep in range(3):
    for two in (False, True):
        res[two].append(run(two, steps))
for two in (False, True):
    ms = sorted(res[two])[1] * 1e3
    print("B=%d  %-28s %.2f ms per forward  %.1f pairs/s" % (B, "two streams, 2 in flight:" if two else "one stream, back to back:", ms, B / ms * 1e3), flush=True)
