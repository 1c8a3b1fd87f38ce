import math, os, sys, statistics, torch
sys.path.insert(0, "/root/repo")
from mickey_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
M = 3878
for (N, K, name) in ((1024, 1024, "proj"), (1024, 4096, "fc2"), (3072, 1024, "qkv"), (4096, 1024, "fc1")):
    a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()
    res = {}
    for mode in (1, 2, 7, -1):
        if mode >= 0:
            ops.gemm_set_tile(mode)
            f = lambda: ops.gemm(a, w, None, out=out)
        else:
            f = lambda: torch.matmul(a, wt, out=out)
        res[mode] = statistics.median([timeit(f, iters=50, warm=5) for _ in range(3)])
    ops.gemm_set_tile(0)
    fl = 2.0 * M * N * K
    print(name, " | ".join("%s %.1f us %4.0f TF" % ({1: "128x128", 2: "64x128 3-stage", 7: "pp64", -1: "hipBLASLt"}[m], t * 1e6, fl / t / 1e12) for m, t in res.items()), flush=True)
