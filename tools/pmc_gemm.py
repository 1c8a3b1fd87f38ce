import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops
dev = torch.device("cuda:0")
M, N, K = 31024, 1024, 4096
a = (torch.randn((M, K), device=dev) * 0.5).bfloat16()
w = (torch.randn((N, K), device=dev) / math.sqrt(K)).bfloat16()
out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
for tile in (2, 7):
    ops.gemm_set_tile(tile)
    for _ in range(3):
        ops.gemm(a, w, None, out=out)
torch.cuda.synchronize()
