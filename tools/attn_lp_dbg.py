"""dev: per-iteration cycle split of the pipelined lean attention kernel (library built with MK_ATTN_LP_DBG=1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops
dev = torch.device("cuda:0")
nimg, heads, ntok, pad = 64, 16, 1939, 1984
q = (torch.randn((nimg, heads, pad, 64), device=dev) * 0.2).bfloat16()
k = torch.randn((nimg, heads, pad, 64), device=dev).bfloat16()
vt = torch.randn((nimg, heads, 64, pad), device=dev).bfloat16()
out = torch.empty((nimg * ntok, heads * 64), device=dev, dtype=torch.bfloat16)
ops.attn_set_mode(6)
for _ in range(3):
    ops.flash_attn(q, k, vt, out, nimg, heads, ntok, pad)
torch.cuda.synchronize()
for w in range(4):
    row = out[128 + 32 * w].view(torch.float32)[:3].cpu().tolist() if False else out[128 + 32 * w, :6].view(torch.float32).cpu().tolist()
    print("wave %d: per iteration  wait+barrier %.0f   head (stage, max tree, rebase check) %.0f   20 slots %.0f cycles" % (w, row[0], row[1], row[2]))
