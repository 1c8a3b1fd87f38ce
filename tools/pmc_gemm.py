"""One GEMM shape under rocprofv3 --pmc (dev tool): GEMM_MODE = mk_gemm_set_tile value, fc2 shape of 16 image pairs."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402

MODE = int(os.environ.get("GEMM_MODE", "0"))   # -1: torch.matmul (hipBLASLt) on the same operands
if MODE >= 0:
    ops.gemm_set_tile(MODE)
dev = torch.device("cuda:0")
M, N, K = 3878 * 16, 1024, 4096
a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
wt = w.t().contiguous()
for _ in range(int(os.environ.get("REPS", "4"))):
    if MODE >= 0:
        ops.gemm(a, w, None, out=out)
    else:
        torch.matmul(a, wt, out=out)
torch.cuda.synchronize()
