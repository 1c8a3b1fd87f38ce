"""Attention kernel variants at the benchmark shape (dev tool): 64 images x 16 heads x 1939 tokens, interleaved medians."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

modes = [int(t) for t in sys.argv[1:]] or [1, 2, 3]
dev = torch.device("cuda:0")
nimg, heads, ntok, pad = int(os.environ.get("NIMG", "64")), 16, 1939, 1984
q = (torch.randn((nimg, heads, pad, 64), device=dev) * 0.2).bfloat16()
k = torch.randn((nimg, heads, pad, 64), device=dev).bfloat16()
vt = torch.randn((nimg, heads, 64, pad), device=dev).bfloat16()
out = torch.empty((nimg * ntok, heads * 64), device=dev, dtype=torch.bfloat16)
ts = {m: [] for m in modes}
for rep in range(5):
    for m in (modes if rep % 2 == 0 else modes[::-1]):
        ops.attn_set_mode(m)
        ts[m].append(timeit(lambda: ops.flash_attn(q, k, vt, out, nimg, heads, ntok, pad), iters=10, warm=2))
ops.attn_set_mode(0)
fl = 4.0 * nimg * heads * ntok * ntok * 64
print("attention %d x 16 x 1939: " % nimg + "  ".join("mode%d %6.1f TF (%.3f ms)" % (m, fl / statistics.median(ts[m]) / 1e12, statistics.median(ts[m]) * 1e3) for m in modes))
