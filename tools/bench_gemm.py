"""GEMM schedule A/B on the encoder shapes (dev tool).  python tools/bench_gemm.py [modes...]
modes: mk_gemm_set_tile values (7 = 8-wave ping-pong); -1 = torch.matmul (hipBLASLt), bare.
Each shape is timed bare (bf16 store) and with its in-forward epilogue (qkv split / LayerScale+residual / bias+GELU).
Repetitions are interleaved over the schedules and the median is reported (clock / power drift on one box is a few %)."""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

modes = [int(t) for t in sys.argv[1:]] or [7, -1]
dev = torch.device("cuda:0")
# NIMG images (64 = the bench batch, 2 = one pair) of WIDTH-wide tokens (1024 = ViT-L, 384 = ViT-S)
nimg, D = int(os.environ.get("NIMG", "64")), int(os.environ.get("WIDTH", "1024"))
ntok, pad, heads = 1939, 1984, D // 64
M = nimg * ntok
for (N, K, name) in ((3 * D, D, "qkv"), (D, D, "proj"), (4 * D, D, "fc1"), (D, 4 * D, "fc2")):
    a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn((N,), device=dev)
    gamma = torch.rand((N,), device=dev)
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    x = torch.randn((M, N), device=dev) if name in ("proj", "fc2") else None
    if name == "qkv":
        q = torch.zeros((nimg, heads, pad, 64), device=dev, dtype=torch.bfloat16)
        k = torch.zeros_like(q)
        vt = torch.zeros((nimg, heads, 64, pad), device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()

    def fused():
        if name == "qkv":
            ops.gemm_qkv(a, w, bias, q, k, vt, nimg, ntok, pad, heads)
        elif name == "fc1":
            ops.gemm(a, w, bias, act=ops.ACT_GELU, out=out)
        else:
            ops.gemm_ls_residual(a, w, bias, gamma, x)

    ts = {(m, f): [] for m in modes for f in (0, 1)}
    for rep in range(5):
        for m in (modes if rep % 2 == 0 else modes[::-1]):
            if m < 0:
                ts[(m, 0)].append(timeit(lambda: torch.matmul(a, wt, out=out), iters=10, warm=2))
                continue
            ops.gemm_set_tile(m)
            ts[(m, 0)].append(timeit(lambda: ops.gemm(a, w, None, out=out), iters=10, warm=2))
            ts[(m, 1)].append(timeit(fused, iters=10, warm=2))
    ops.gemm_set_tile(0)
    fl = 2.0 * M * N * K
    line = "M=%6d D=%4d %-4s " % (M, D, name)
    for m in modes:
        line += " | %s bare %6.1f" % ("hipBLASLt" if m < 0 else "mode%d" % m, fl / statistics.median(ts[(m, 0)]) / 1e12)
        if m >= 0:
            line += " fused %6.1f" % (fl / statistics.median(ts[(m, 1)]) / 1e12)
    print(line + " TF", flush=True)
