"""GEMM timing ablations of the K=32 ring kernel (dev tool).  Needs a library built with -DMK_GEMM_ABLATIONS: those
instantiations are not in the default build."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
M, N, K = 124096, 1024, 4096
a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
names = {2: "t2", 3: "pp", 18: "pp dma from hot addresses", 19: "pp dma 8-line pieces (cold)", 20: "pp dma 8-line pieces (hot)", 21: "pp plain global loads instead of dma (no lds write)", 11: "pp -dma", 12: "pp -ldsread", 13: "pp -dma -ldsread", 14: "pp -barrier", 15: "pp -dma -barrier",
         16: "pp -ldsread -barrier", 17: "pp mfma only"}
for mode, nm in names.items():
    ops.gemm_set_tile(mode)
    t = timeit(lambda: ops.gemm(a, w, None, out=out), iters=10)
    print("%-24s %.3f ms  %.0f TF" % (nm, t * 1e3, 2.0 * M * N * K / t / 1e12), flush=True)
