"""Tile ORDER A/B of the 256x256 GEMM on the encoder shapes (dev tool): bands of 8 m-tiles walked n-major (mk_gemm_set_tile 408)
against groups of g n-tiles walked m-major (464 + g).  python tools/bench_gemm_order.py [orders...]
ORDER_ONLY=<order> runs each shape's fused form a few times under that one order (for a rocprofv3 --pmc FETCH_SIZE pass)."""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

orders = [int(t) for t in sys.argv[1:]] or [408, 468, 470, 472]
only = os.environ.get("ORDER_ONLY")
dev = torch.device("cuda:0")
nimg, D = int(os.environ.get("NIMG", "64")), 1024
ntok, pad, heads = 1939, 1984, D // 64
M = nimg * ntok
ops.gemm_set_tile(7)
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

    def fused():
        if name == "qkv":
            ops.gemm_qkv(a, w, bias, q, k, vt, nimg, ntok, pad, heads)
        elif name == "fc1":
            ops.gemm(a, w, bias, act=ops.ACT_GELU, out=out)
        else:
            ops.gemm_ls_residual(a, w, bias, gamma, x)

    if only:
        ops.gemm_set_tile(int(only))
        for _ in range(3):
            fused()
        torch.cuda.synchronize()
        continue
    ts = {o: [] for o in orders}
    for rep in range(5):
        for o in (orders if rep % 2 == 0 else orders[::-1]):
            ops.gemm_set_tile(o)
            ts[o].append(timeit(fused, iters=10, warm=2))
    fl = 2.0 * M * N * K
    print("M=%6d %-4s fused " % (M, name) + " | ".join("order %d: %6.1f TF" % (o, fl / statistics.median(ts[o]) / 1e12) for o in orders), flush=True)
ops.gemm_set_tile(464)
ops.gemm_set_tile(0)
