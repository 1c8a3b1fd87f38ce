"""Tile order of the 256x256 kernel for the heads' CONVOLUTIONS (dev tool): the automatic rule covers dense GEMMs only; here the
whole forward runs with the order forced (mk_gemm_set_tile 400 = automatic, 468 = n-fastest groups of 4 for every launch) and the
conv_gemm stage is read from the HIP events.  (The forced order also applies to qkv / fc1, which it slows: read the conv row.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mickey_amd import ops, synthetic as syn  # noqa: E402
from mickey_amd.config import default_cfg  # noqa: E402
from mickey_amd.model import MickeyRelativePose  # noqa: E402

dev = torch.device("cuda:0")
prof = bench.StageProfiler()
prof.wrap(ops)
cfg = default_cfg()
cfg["AMD"]["ENCODER_DTYPE"] = "bf16"
m = MickeyRelativePose(cfg)
m.load_state_dict(syn.mickey_state_dict(cfg, seed=0, arch="vit_large"))
m = m.to(dev)
data = {k: v.to(dev) for k, v in syn.synthetic_batch(B=32, H=720, W=540, seed=1234).items()}
for _ in range(2):
    m(dict(data))
res = {}
for rep in range(3):
    for order in (400, 468):
        ops.gemm_set_tile(order)
        m(dict(data))
        torch.cuda.synchronize()
        prof.records = []
        prof.on = True
        for _ in range(4):
            m(dict(data))
        torch.cuda.synchronize()
        prof.on = False
        st, _ = prof.summary(4)
        for s in st:
            res.setdefault((order, s["stage"]), []).append(s["ms_per_step"])
ops.gemm_set_tile(400)
for stage in ("conv_gemm", "encoder_gemm"):
    print("%-13s " % stage + "   ".join("order %d: %.3f ms/step (median of 3)" % (o, sorted(res[(o, stage)])[1]) for o in (400, 468)), flush=True)
