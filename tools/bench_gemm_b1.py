"""One image pair (M = 3878): the four encoder GEMM shapes bare on every tile form (mk_gemm_set_tile 1 / 2 / 7; round 6 also measured a 4-stage 128x128 form, since removed: profiles/r06k_gemm_b1.txt) next to hipBLASLt,
and the two producers (proj, fc2) in the form the forward runs them (LayerScale + residual on the split stream + row statistics)."""
import math, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
M = 3878
NAMES = {0: "auto", 1: "128x128 2-stage", 2: "64x128 3-stage", 7: "pp64", -1: "hipBLASLt"}
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
    print("M=%d bare  %-4s" % (M, name), " | ".join("%s %.1f us %4.0f TF" % (NAMES[m], t * 1e6, fl / t / 1e12) for m, t in res.items()), flush=True)
    if N != 1024:
        continue
    hi = torch.randn((M, N), device=dev).bfloat16()
    lo = torch.zeros_like(hi)
    st = torch.empty((M, N // 64, 2), device=dev)
    b, gamma, shift = torch.randn((N,), device=dev), torch.rand((N,), device=dev), torch.randn((M,), device=dev) * 0.1
    res = {}
    for mode in (1, 2, 7, 0):
        ops.gemm_set_tile(mode)
        f = lambda: ops.gemm_ls_residual_ln(a, w, b, gamma, hi, lo, st, shift=shift)
        res[mode] = statistics.median([timeit(f, iters=50, warm=5) for _ in range(3)])
    ops.gemm_set_tile(0)
    print("M=%d fused %-4s" % (M, name), " | ".join("%s %.1f us %4.0f TF" % (NAMES[m], t * 1e6, fl / t / 1e12) for m, t in res.items()), flush=True)
