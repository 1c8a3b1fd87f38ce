"""Outer sampler (mk_exprace_topk, 20 x 2048 of 1938^2 cells per pair) A/B of the two on-device generators (dev tool):
mode 0 = candidates by geometric skipping, mode 1 = every (row, cell) tested behind the 6-bit pre-filter.
Inputs: the bench's near-uniform final_scores (dual-softmax of random descriptors) and a peaked one (3000 dominant cells)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
B, n = int(os.environ.get("B", "32")), 1938
g = torch.Generator(device=dev).manual_seed(0)
flat = torch.rand((B, n * n), device=dev, generator=g) * 2e-7 + 1e-7
peaked = flat.clone()
for b in range(B):
    sel = torch.randperm(n * n, device=dev, generator=g)[:3000]
    peaked[b, sel] = torch.rand(3000, device=dev, generator=g) * 0.5 + 0.05
for name, p in (("near-uniform", flat), ("peaked (3000 dominant cells)", peaked)):
    for mode in (0, 1):
        ops.exprace_set_mode(mode)
        work = ops.exprace_work(B, 20, 2048, n * n, dev)   # zeroed once, self-cleaning afterwards: what the model's workspace holds
        idx, cnt = ops.exprace_topk(p, 20, 2048, seed=1, offset=0, work=work)
        assert int(cnt.min()) == 2048
        t = timeit(lambda: ops.exprace_topk(p, 20, 2048, seed=1, offset=3, work=work), iters=20, warm=3)
        print("B=%d %-30s mode %d (%s): %.3f ms" % (B, name, mode, "skip" if mode == 0 else "pre-filter", t * 1e3), flush=True)
ops.exprace_set_mode(0)
