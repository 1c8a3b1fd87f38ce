"""Race screen of the persistent tile loop (dev tool; the guide: "sync-structure edits are NEW templates: multi-run race screen"):
for a sweep of row counts (ragged last tiles, different numbers of tiles per workgroup) every folded-LN encoder GEMM form is run
REPS times persistent and compared bit for bit with ONE one-tile-per-workgroup run of the same launch; other kernels (a big
matmul, an attention launch) are interleaved at random so that clocks, L2 contents and the memory system differ from run to run."""
import math
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "12"))
lp = torch.bfloat16
D, heads, ntok, pad = 1024, 16, 1939, 1984
rnd = random.Random(3)
junk_a = torch.randn((8192, 4096), device=dev).to(lp)
junk_b = torch.randn((4096, 4096), device=dev).to(lp)
ops.gemm_set_tile(7)
bad = total = 0
for nimg in (10, 14, 20, 33, 64):
    M = nimg * ntok
    g = torch.Generator(device="cuda").manual_seed(nimg)
    rn = lambda *s, sc=1.0: torch.randn(s, device=dev, generator=g) * sc  # noqa: E731
    xf = rn(M, D) * 2 + 0.5
    xh = xf.to(lp)
    xl0 = (xf - xh.float()).to(lp)
    stats = torch.stack([xf.double().reshape(M, 16, 64).sum(-1), (xf.double() ** 2).reshape(M, 16, 64).sum(-1)], -1).float()
    a4 = rn(M, 4 * D, sc=0.5).to(lp)
    forms = {}
    for name, N, K in (("qkv", 3 * D, D), ("fc1", 4 * D, D), ("proj", D, D), ("fc2", D, 4 * D)):
        w = rn(N, K, sc=1 / math.sqrt(K)).to(lp)
        b, c, gam, sh = rn(N, sc=0.1), rn(N), torch.rand((N,), device=dev, generator=g), rn(M, sc=0.3)
        if name == "qkv":
            def run(w=w, b=b, c=c):
                q = torch.zeros((nimg, heads, pad, 64), device=dev, dtype=lp)
                k, vt, so = torch.zeros_like(q), torch.zeros((nimg, heads, 64, pad), device=dev, dtype=lp), torch.zeros((M,), device=dev)
                ops.gemm_qkv_ln(xh, w, b, c, stats, 1e-6, q, k, vt, nimg, ntok, pad, heads, shift_out=so)
                return [q, k, vt, so]
        elif name == "fc1":
            def run(w=w, b=b, c=c):
                so = torch.zeros((M,), device=dev)
                return [ops.gemm_ln(xh, w, b, c, stats, 1e-6, act=ops.ACT_GELU, shift_out=so), so]
        else:
            def run(w=w, b=b, gam=gam, sh=sh, K=K):
                hi, lo, st = xh.clone(), xl0.clone(), torch.zeros((M, 16, 2), device=dev)
                ops.gemm_ls_residual_ln(a4[:, :K] if K == 4 * D else a4[:, :D].contiguous(), w, b, gam, hi, lo, st, shift=sh)
                return [hi, lo, st]
        forms[name] = run
    for name, run in forms.items():
        ops.gemm_set_tile(601)
        ref = run()
        ops.gemm_set_tile(600)
        nbad = 0
        for rep in range(REPS):
            if rnd.random() < 0.5:
                torch.matmul(junk_a[: rnd.choice((1024, 4096, 8192))], junk_b)
            out = run()
            nbad += int(not all(torch.equal(x, y) for x, y in zip(out, ref)))
        total += REPS
        bad += nbad
        print("M=%6d (%3d m-tiles) %-4s: %d / %d persistent runs differ from the one-tile-per-workgroup result" % (M, (M + 255) // 256, name, nbad, REPS), flush=True)
print("TOTAL: %d / %d differ" % (bad, total))
sys.exit(1 if bad else 0)
