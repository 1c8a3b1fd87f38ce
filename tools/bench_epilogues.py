import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops
from tools.bench_kernels import timeit
dev = torch.device("cuda:0")
nimg = 64; ntok = 1939; M = nimg * ntok; D = 1024; pad = 1984
def mk(m, k): return (torch.randn((m, k), device=dev) * 0.5).bfloat16()
def wt(n, k): return (torch.randn((n, k), device=dev) / math.sqrt(k)).bfloat16()
y = mk(M, D); hid = mk(M, 4 * D)
wq, wp, w1, w2 = wt(3 * D, D), wt(D, D), wt(4 * D, D), wt(D, 4 * D)
bq, bp, b1, b2 = (torch.randn(n, device=dev) for n in (3 * D, D, 4 * D, D))
g = torch.rand(D, device=dev); x = torch.randn((M, D), device=dev)
q = torch.zeros((nimg, 16, pad, 64), device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q)
vt = torch.zeros((nimg, 16, 64, pad), device=dev, dtype=torch.bfloat16)
out3 = torch.empty((M, 3 * D), device=dev, dtype=torch.bfloat16); out4 = torch.empty((M, 4 * D), device=dev, dtype=torch.bfloat16)
out1 = torch.empty((M, D), device=dev, dtype=torch.bfloat16)
for tile, stag in ((7, 0), (7, 0)):
    ops.gemm_set_tile(100 + stag)
    ops.gemm_set_tile(tile)
    r = {}
    r["qkv plain"] = timeit(lambda: ops.gemm(y, wq, None, out=out3))
    r["qkv split epi"] = timeit(lambda: ops.gemm_qkv(y, wq, bq, q, k, vt, nimg, ntok, pad, 16))
    r["fc1 plain"] = timeit(lambda: ops.gemm(y, w1, None, out=out4))
    r["fc1 bias"] = timeit(lambda: ops.gemm(y, w1, b1, out=out4))
    r["fc1 bias+gelu"] = timeit(lambda: ops.gemm(y, w1, b1, act=ops.ACT_GELU, out=out4))
    r["fc2 plain"] = timeit(lambda: ops.gemm(hid, w2, None, out=out1))
    r["fc2 ls_resid"] = timeit(lambda: ops.gemm_ls_residual(hid, w2, b2, g, x))
    r["proj plain"] = timeit(lambda: ops.gemm(y, wp, None, out=out1))
    r["proj ls_resid"] = timeit(lambda: ops.gemm_ls_residual(y, wp, bp, g, x))
    print("tile", tile, "stagger", stag, {k_: "%.3f ms" % (v * 1e3) for k_, v in r.items()}, flush=True)
