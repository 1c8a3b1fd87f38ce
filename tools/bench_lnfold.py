"""Per-GEMM cost of the folded LayerNorm at the benchmarked shapes (M = 64 x 1939): each encoder GEMM in its plain form
and in its *_ln form (producer: split residual stream + row statistics; consumer: + row parameters + correction), interleaved."""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
nimg, ntok, pad, heads = 64, 1939, 1984, 16
M, D = nimg * ntok, 1024
lp = torch.bfloat16
xs = (torch.randn((M, D), device=dev)).to(lp)
xl = torch.zeros_like(xs)
stats = torch.rand((M, D // 64, 2), device=dev) * 64 + 64
x = torch.randn((M, D), device=dev)
hid = (torch.randn((M, 4 * D), device=dev) * 0.5).to(lp)
att = (torch.randn((M, D), device=dev) * 0.5).to(lp)
q = torch.zeros((nimg, heads, pad, 64), device=dev, dtype=lp)
k = torch.zeros_like(q)
vt = torch.zeros((nimg, heads, 64, pad), device=dev, dtype=lp)


def mk(N, K):
    return ((torch.randn((N, K), device=dev) / math.sqrt(K)).to(lp), torch.randn((N,), device=dev) * 0.1,
            torch.randn((N,), device=dev), torch.rand((N,), device=dev) * 1e-3)


wq, bq, cq, _ = mk(3 * D, D)
w1, b1, c1, _ = mk(4 * D, D)
wp, bp, _, gp = mk(D, D)
w2, b2, _, g2 = mk(D, 4 * D)
out1 = torch.empty((M, 4 * D), device=dev, dtype=lp)
cases = {
    "qkv": (lambda: ops.gemm_qkv(xs, wq, bq, q, k, vt, nimg, ntok, pad, heads),
            lambda: ops.gemm_qkv_ln(xs, wq, bq, cq, stats, 1e-6, q, k, vt, nimg, ntok, pad, heads), 2.0 * M * 3 * D * D),
    "fc1": (lambda: ops.gemm(xs, w1, b1, act=ops.ACT_GELU, out=out1),
            lambda: ops.gemm_ln(xs, w1, b1, c1, stats, 1e-6, act=ops.ACT_GELU, out=out1), 2.0 * M * 4 * D * D),
    "proj": (lambda: ops.gemm_ls_residual(att, wp, bp, gp, x),
             lambda: ops.gemm_ls_residual_ln(att, wp, bp, gp, xs, xl, stats), 2.0 * M * D * D),
    "fc2": (lambda: ops.gemm_ls_residual(hid, w2, b2, g2, x),
            lambda: ops.gemm_ls_residual_ln(hid, w2, b2, g2, xs, xl, stats), 2.0 * M * D * 4 * D),
}
for name, (plain, fold, fl) in cases.items():
    tp, tf = [], []
    for rep in range(5):
        for which in ((0, 1) if rep % 2 == 0 else (1, 0)):
            (tf if which else tp).append(timeit(fold if which else plain, iters=10, warm=2))
    a, b = statistics.median(tp), statistics.median(tf)
    print("%-4s plain %.3f ms (%4.0f TF) | folded-LN form %.3f ms (%4.0f TF) | +%.3f ms" % (name, a * 1e3, fl / a / 1e12, b * 1e3, fl / b / 1e12, (b - a) * 1e3), flush=True)
