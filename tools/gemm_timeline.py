"""Per-tile timeline of the full-line ping-pong GEMM (dev tool): where does a tile's time go?
    python tools/gemm_timeline.py [tile_mode ...]
Prints, per schedule and shape: K-loop time, prologue (entry -> first stage landed), epilogue issue time, and the gap
between a workgroup's end and the next workgroup's entry on the same CU (hardware relaunch)."""
import math
import os
import statistics
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
modes = [int(t) for t in sys.argv[1:]] or [7, 5]
M = 3878 * 32
for (N, K, name) in ((1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")):
    a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    ntiles = ((M + 255) // 256) * ((N + 255) // 256)
    for mode in modes:
        ops.gemm_set_tile(mode)
        buf = torch.zeros((ntiles * 2, 6), device=dev, dtype=torch.int64)
        for _ in range(3):
            ops.gemm(a, w, None, out=out)
        torch.cuda.synchronize()
        ops.gemm_debug_timeline(buf)
        ops.gemm(a, w, None, out=out)
        torch.cuda.synchronize()
        ops.gemm_debug_timeline(None)
        d = buf.cpu().numpy().reshape(ntiles, 2, 6)
        t0 = d[:, :, 0].min()
        us = lambda x: x / 100.0  # 100 MHz ticks -> us
        ent, land, loop, end = (us(d[:, :, i] - t0) for i in range(4))
        hw = d[:, 0, 4]
        cu = (d[:, 0, 5] & 0xF) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xF)
        bycu = defaultdict(list)
        for t in range(ntiles):
            bycu[int(cu[t])].append((ent[t].min(), end[t].max()))
        gaps = []
        for v in bycu.values():
            v.sort()
            gaps += [b[0] - a_[1] for a_, b in zip(v[:-1], v[1:])]
        med = statistics.median
        print("%-4s mode %d: total %.0f us | CUs %d | prologue %.2f | K loop row0 %.2f row1 %.2f | epilogue row0 %.2f row1 %.2f | "
              "tile %.2f | relaunch gap med %.2f p90 %.2f us" % (
                  name, mode, end.max(), len(bycu), med((land - ent).ravel()), med(loop[:, 0] - land[:, 0]), med(loop[:, 1] - land[:, 1]),
                  med(end[:, 0] - loop[:, 0]), med(end[:, 1] - loop[:, 1]), med(end.max(1) - ent.min(1)),
                  med(gaps) if gaps else -1, sorted(gaps)[int(len(gaps) * 0.9)] if gaps else -1), flush=True)
    ops.gemm_set_tile(0)
