"""The encoder GEMMs in the forms the forward launches (folded LayerNorm: qkv / fc1 consumers, proj / fc2 producers on the split
residual stream) at the benchmarked shapes (M = 64 x 1939), one tile per workgroup (mk_gemm_set_tile 601) against the persistent
tile loop (600), interleaved medians; torch.matmul (hipBLASLt, bare: no epilogue) beside them."""
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
nimg, ntok, pad, heads = int(os.environ.get("NIMG", "64")), 1939, 1984, 16
M, D = nimg * ntok, 1024
lp = torch.bfloat16
xs = (torch.randn((M, D), device=dev)).to(lp)
xl = torch.zeros_like(xs)
stats = torch.rand((M, D // 64, 2), device=dev) * 64 + 64
shift = torch.randn((M,), device=dev) * 0.1
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
sh_out = torch.empty((M,), device=dev)
cases = {
    "qkv": (lambda: ops.gemm_qkv_ln(xs, wq, bq, cq, stats, 1e-6, q, k, vt, nimg, ntok, pad, heads, shift_out=sh_out), 3 * D, D, xs, wq),
    "proj": (lambda: ops.gemm_ls_residual_ln(att, wp, bp, gp, xs, xl, stats, shift=shift), D, D, att, wp),
    "fc1": (lambda: ops.gemm_ln(xs, w1, b1, c1, stats, 1e-6, act=ops.ACT_GELU, out=out1, shift_out=sh_out), 4 * D, D, xs, w1),
    "fc2": (lambda: ops.gemm_ls_residual_ln(hid, w2, b2, g2, xs, xl, stats, shift=shift), D, 4 * D, hid, w2),
}
ops.gemm_set_tile(7)
tot = {0: 0.0, 1: 0.0}
for name, (fn, N, K, a, w) in cases.items():
    fl = 2.0 * M * N * K
    wt = w.t().contiguous()
    ts = {0: [], 1: [], 2: []}
    for rep in range(7):
        order = (0, 1, 2) if rep % 2 == 0 else (2, 1, 0)
        for which in order:
            if which == 2:
                ts[2].append(timeit(lambda: torch.matmul(a, wt), iters=10, warm=2))
            else:
                ops.gemm_set_tile(601 - which)
                ts[which].append(timeit(fn, iters=10, warm=2))
    off, on, bl = (statistics.median(ts[i]) for i in (0, 1, 2))
    tot[0] += off
    tot[1] += on
    print("M=%d %-4s N=%4d K=%4d | one tile per workgroup %.3f ms (%4.0f TF) | persistent %.3f ms (%4.0f TF) %+5.1f %% | hipBLASLt bare %.3f ms (%4.0f TF) | persistent / bare %.2f"
          % (M, name, N, K, off * 1e3, fl / off / 1e12, on * 1e3, fl / on / 1e12, (off / on - 1) * 100, bl * 1e3, fl / bl / 1e12, bl / on), flush=True)
ops.gemm_set_tile(602)   # back to the default (producers only)
print("M=%d sum of the four (one encoder block's linears): %.3f -> %.3f ms (%+.1f %%); x 24 blocks = %.1f -> %.1f ms per step"
      % (M, tot[0] * 1e3, tot[1] * 1e3, (tot[0] / tot[1] - 1) * 100, tot[0] * 24e3, tot[1] * 24e3))
