"""GEMM tile-schedule A/B on the encoder shapes (dev tool).  python tools/bench_gemm.py [tiles...]
Repetitions are interleaved over the schedules and the median is reported (clock / power drift on one box is a few %)."""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

tiles = [int(t) for t in sys.argv[1:]] or [2, 3, 5]   # values >= 400 set the band height (400 + b) of schedule 7
dev = torch.device("cuda:0")
for M in (3878 * 32,):
    for (N, K, name) in ((3072, 1024, "qkv"), (1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")):
        a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
        w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
        out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        ts = {t: [] for t in tiles}
        for rep in range(5):
            for tile in (tiles if rep % 2 == 0 else tiles[::-1]):
                if tile >= 400:
                    ops.gemm_set_tile(tile)
                    ops.gemm_set_tile(7)
                else:
                    ops.gemm_set_tile(408)
                    ops.gemm_set_tile(tile)
                ts[tile].append(timeit(lambda: ops.gemm(a, w, None, out=out), iters=10, warm=2))
        ops.gemm_set_tile(0)
        print("M=%6d %-4s " % (M, name) + "  ".join("tile%d %7.1f TF" % (t, 2.0 * M * N * K / statistics.median(ts[t]) / 1e12) for t in tiles), flush=True)
