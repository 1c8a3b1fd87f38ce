"""One GEMM shape under rocprofv3 --pmc (dev tool): GEMM_MODE = mk_gemm_set_tile value; GEMM_SHAPE = fc2 | fc1 | long (K = 9216, N = 512) of 16 image pairs, or
conv = the first 3x3 convolution of the heads on the same pairs."""
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
SHAPE = os.environ.get("GEMM_SHAPE", "fc2")   # fc2 | fc1 | conv (rb0's first conv: 4 heads x 1024 -> 512 channels, 51 x 38 grids)
REPS = int(os.environ.get("REPS", "4"))
if SHAPE == "conv":
    nimg, H, W, C1, Cout, G = 32, 51, 38, 1024, 512, 4
    R = ops.bordered_rows(nimg, H, W)
    x = torch.zeros((R, C1), device=dev, dtype=torch.bfloat16)
    x[ops.bordered_index(nimg, H, W, dev)] = (torch.randn((nimg * H * W, C1), device=dev) * 0.5).bfloat16()
    w = (torch.randn((G, Cout, 9 * C1), device=dev) / math.sqrt(9 * C1)).bfloat16()
    bias = torch.zeros((G, Cout), device=dev)
    out = torch.zeros((G, R, Cout), device=dev, dtype=torch.bfloat16)
    for _ in range(REPS):
        ops.conv3x3(x, C1, w, bias, out, Cout, G, nimg, H, W, act=ops.ACT_RELU, stride_in1=0, stride_w=Cout * 9 * C1,
                    stride_bias=Cout, stride_out=R * Cout, out_bordered=True)
else:
    # long: the conv's GEMM shape (K = 9216, 512 output channels) as a dense problem
    M, N, K = {"fc2": (3878 * 16, 1024, 4096), "fc1": (3878 * 16, 4096, 1024), "long": (3876 * 16, 512, 9216)}[SHAPE]
    a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
    out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()
    for _ in range(REPS):
        if MODE >= 0:
            ops.gemm(a, w, None, out=out)
        else:
            torch.matmul(a, wt, out=out)
torch.cuda.synchronize()
