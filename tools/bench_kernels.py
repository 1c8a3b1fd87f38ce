"""Micro-benchmarks of the dominant kernels on the GPU box (dev tool, not part of the product).
    python tools/bench_kernels.py [--json gpurun_out/kernels.json]
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = []
    for M in (3878, 3878 * 8, 3878 * 32):
        for (N, K, name) in ((3072, 1024, "qkv"), (1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")):
            a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
            w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
            out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
            for tile in (1, 7):
                ops.gemm_set_tile(tile)
                t = timeit(lambda: ops.gemm(a, w, None, out=out))
                tf = 2.0 * M * N * K / t / 1e12
                res.append({"kernel": "gemm_" + name, "tile": tile, "M": M, "N": N, "K": K, "ms": t * 1e3, "TFLOPs": tf})
                print(res[-1], flush=True)
            ops.gemm_set_tile(0)
    for nimg in (2, 16, 64):
        heads, ntok, pad = 16, 1939, 1984
        q = (torch.randn((nimg, heads, pad, 64), device=dev) * 0.2).bfloat16()
        k = torch.randn((nimg, heads, pad, 64), device=dev).bfloat16()
        vt = torch.randn((nimg, heads, 64, pad), device=dev).bfloat16()
        out = torch.empty((nimg * ntok, heads * 64), device=dev, dtype=torch.bfloat16)
        for mode in (1, 2, 3):
            ops.attn_set_mode(mode)
            t = timeit(lambda: ops.flash_attn(q, k, vt, out, nimg, heads, ntok, pad))
            tf = 4.0 * nimg * heads * ntok * ntok * 64 / t / 1e12
            res.append({"kernel": "flash_attn", "mode": mode, "nimg": nimg, "ms": t * 1e3, "TFLOPs": tf})
            print(res[-1], flush=True)
        ops.attn_set_mode(0)
    for M in (3878, 3878 * 8):
        x = torch.randn((M, 1024), device=dev)
        w = torch.ones(1024, device=dev)
        out = torch.empty((M, 1024), device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.layernorm(x, w, w, 1e-6, out=out))
        res.append({"kernel": "layernorm", "M": M, "ms": t * 1e3, "GBs": M * 1024 * 6 / t / 1e9})
        print(res[-1], flush=True)
    B, n = 4, 1938
    d0 = torch.nn.functional.normalize(torch.randn((B, 128, n), device=dev), dim=1)
    d1 = torch.nn.functional.normalize(torch.randn((B, 128, n), device=dev), dim=1)
    s0 = torch.rand((B, 1, n), device=dev)
    t = timeit(lambda: ops.dual_softmax(d0, d1, s0, s0, 0.1, 1.0), iters=10)
    res.append({"kernel": "dual_softmax(3 outputs)", "B": B, "ms": t * 1e3, "GBs": B * (3 * n * n * 4 + 2 * 128 * n * 4) / t / 1e9})
    print(res[-1], flush=True)
    if args.json:
        os.makedirs(os.path.dirname(args.json), exist_ok=True)
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
