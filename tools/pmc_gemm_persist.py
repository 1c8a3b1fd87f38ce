"""One encoder GEMM in its folded-LayerNorm form under rocprofv3 --pmc (dev tool, LABNOTES R5.2): GEMM_SHAPE = qkv | proj | fc1 | fc2 at
M = 64 x 1939, PERSIST = 1 / 0 (mk_gemm_set_tile 600 / 601: persistent tile loop / one tile per workgroup), REPS launches."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPE, REPS = os.environ.get("GEMM_SHAPE", "fc1"), int(os.environ.get("REPS", "12"))
ops.gemm_set_tile(7)
ops.gemm_set_tile(600 if int(os.environ.get("PERSIST", "1")) else 601)
nimg, ntok, pad, heads, D = 64, 1939, 1984, 16, 1024
M, lp = nimg * ntok, torch.bfloat16
N, K = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}[SHAPE]
a = (torch.randn((M, K), device=dev) * 0.5).to(lp)
w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(lp)
b, c = torch.randn((N,), device=dev) * 0.1, torch.randn((N,), device=dev)
stats = torch.rand((M, D // 64, 2), device=dev) * 64 + 64
if SHAPE in ("proj", "fc2"):
    xh, xl = torch.randn((M, D), device=dev).to(lp), torch.zeros((M, D), device=dev, dtype=lp)
    gam, sh = torch.rand((N,), device=dev) * 1e-3, torch.randn((M,), device=dev) * 0.1
    fn = lambda: ops.gemm_ls_residual_ln(a, w, b, gam, xh, xl, stats, shift=sh)  # noqa: E731
elif SHAPE == "fc1":
    out = torch.empty((M, N), device=dev, dtype=lp)
    fn = lambda: ops.gemm_ln(a, w, b, c, stats, 1e-6, act=ops.ACT_GELU, out=out)  # noqa: E731
else:
    q = torch.zeros((nimg, heads, pad, 64), device=dev, dtype=lp)
    k, vt = torch.zeros_like(q), torch.zeros((nimg, heads, 64, pad), device=dev, dtype=lp)
    fn = lambda: ops.gemm_qkv_ln(a, w, b, c, stats, 1e-6, q, k, vt, nimg, ntok, pad, heads)  # noqa: E731
for _ in range(REPS):
    fn()
torch.cuda.synchronize()
