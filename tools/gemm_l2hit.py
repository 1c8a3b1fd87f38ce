"""Is the GEMM K loop bound by DMA latency (L2 misses) or by LDS / issue?  (dev tool)
Grouped GEMM, 2048 groups of one 256x256xK tile each: (a) all groups share one A and one W panel (pure L2 hits),
(b) every group has its own panels (every DMA line is an L2 miss).  Same flops, same schedule."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
G, M, N = 2048, 256, 256
for K in (1024, 4096):
    a = (torch.randn((G, M, K), device=dev) * 0.5).bfloat16()
    w = (torch.randn((G, N, K), device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty((G, M, N), device=dev, dtype=torch.bfloat16)
    for mode in (7, 10):
        ops.gemm_set_tile(mode)
        for name, sa, sw in (("shared panels (L2 hits)", 0, 0), ("own panels (misses)", M * K, N * K)):
            t = timeit(lambda: ops.gemm_grouped(a, w, None, out, G, M, N, K, K, K, N, sa, sw, 0, M * N), iters=10, warm=2)
            print("K=%d mode %d %-26s %.3f ms  %.0f TF  %.2f us/stage(64)" % (K, mode, name, t * 1e3, 2.0 * G * M * N * K / t / 1e12,
                                                                  t * 1e6 / (G / 256) / (K / 64)), flush=True)
    ops.gemm_set_tile(0)
