import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops
dev = torch.device("cuda:0")
nimg, heads, ntok, pad = int(os.environ.get("NIMG", "16")), 16, 1939, 1984
q = (torch.randn((nimg, heads, pad, 64), device=dev) * 0.2).bfloat16()
k = torch.randn((nimg, heads, pad, 64), device=dev).bfloat16()
vt = torch.randn((nimg, heads, 64, pad), device=dev).bfloat16()
out = torch.empty((nimg * ntok, heads * 64), device=dev, dtype=torch.bfloat16)
ops.attn_set_mode(int(os.environ.get("ATTN_MODE", "0")))
for _ in range(int(os.environ.get("REPS", "3"))):
    ops.flash_attn(q, k, vt, out, nimg, heads, ntok, pad)
torch.cuda.synchronize()
