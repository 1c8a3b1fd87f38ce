"""Where do the hand-written kernels stand against the vendor libraries on the same box?  (dev tool, not product code)
torch.matmul (hipBLASLt / rocBLAS) on the encoder GEMM shapes, F.scaled_dot_product_attention on the attention shape."""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
M = 3878 * 32
for (N, K, name) in ((3072, 1024, "qkv"), (1024, 1024, "proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2")):
    a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    t_mine = timeit(lambda: ops.gemm(a, w, None, out=out), iters=10, warm=3)
    t_lib = timeit(lambda: torch.matmul(a, w.t(), out=out), iters=10, warm=3)
    fl = 2.0 * M * N * K
    print("gemm %-4s M=%d N=%d K=%d: mickey_amd %.0f TFLOP/s, torch.matmul %.0f TFLOP/s" % (name, M, N, K, fl / t_mine / 1e12, fl / t_lib / 1e12), flush=True)
nimg, heads, ntok, pad = 64, 16, 1939, 1984
q = (torch.randn((nimg, heads, pad, 64), device=dev) * 0.2).bfloat16()
k = torch.randn((nimg, heads, pad, 64), device=dev).bfloat16()
v = torch.randn((nimg, heads, pad, 64), device=dev).bfloat16()
vt = v.transpose(-1, -2).contiguous()
out = torch.empty((nimg * ntok, heads * 64), device=dev, dtype=torch.bfloat16)
t_mine = timeit(lambda: ops.flash_attn(q, k, vt, out, nimg, heads, ntok, pad), iters=10, warm=3)
fl = 4.0 * nimg * heads * ntok * ntok * 64
qs, ks, vs = q[:, :, :ntok], k[:, :, :ntok], v[:, :, :ntok]
try:
    t_lib = timeit(lambda: F.scaled_dot_product_attention(qs, ks, vs), iters=5, warm=2)
    print("attention %d x %d heads x %d tokens: mickey_amd %.0f TFLOP/s, F.scaled_dot_product_attention %.0f TFLOP/s" % (nimg, heads, ntok, fl / t_mine / 1e12, fl / t_lib / 1e12))
except Exception as e:  # noqa: BLE001
    print("attention: mickey_amd %.0f TFLOP/s, SDPA failed: %s" % (fl / t_mine / 1e12, e))
x = torch.randn((M, 1024), device=dev)
wln = torch.ones(1024, device=dev)
o = torch.empty((M, 1024), device=dev, dtype=torch.bfloat16)
t_mine = timeit(lambda: ops.layernorm(x, wln, wln, 1e-6, out=o), iters=10, warm=3)
t_lib = timeit(lambda: F.layer_norm(x, (1024,), wln, wln, 1e-6).to(torch.bfloat16), iters=10, warm=3)
print("layernorm fp32 -> bf16, %d rows: mickey_amd %.2f TB/s, torch (layer_norm + cast) %.2f TB/s of the same 6 B/element" % (M, M * 1024 * 6 / t_mine / 1e12, M * 1024 * 6 / t_lib / 1e12))
