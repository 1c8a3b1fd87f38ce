"""Matcher stage alone at the bench shape (32 pairs, n = 1938): the exact fp32-MFMA path and the split-fp16 path of the dual softmax,
full (scores + kp_scores + final_scores) and lean (final_scores only), HIP-event medians; then the Sinkhorn variant at the
config-#5 shape (8 pairs, n = 4641) with the pairs iterated batch-wide (round 3) / in groups that fit the Infinity Cache.
Run under `rocprofv3 --kernel-trace --stats --output-format csv` for the per-kernel split."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mickey_amd import ops  # noqa: E402


def timed(fn, reps=15):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "dual"):
        B, n = 32, 1938
        d0 = torch.nn.functional.normalize(torch.randn((B, 128, n), generator=g), dim=1).to(dev)
        d1 = torch.nn.functional.normalize(torch.randn((B, 128, n), generator=g), dim=1).to(dev)
        s0 = (torch.rand((B, 1, n), generator=g) / n).to(dev)
        s1 = (torch.rand((B, 1, n), generator=g) / n).to(dev)
        for split, chunks in ((False, 0), (True, 8), (True, 16), (True, 0)):
            ops.dual_softmax_set_chunks(chunks)
            for lean in (False, True):
                t = timed(lambda: ops.dual_softmax(d0, d1, s0, s1, 0.1, 1.0, want_scores=not lean, want_kp=not lean, split=split))
                nbytes = 4.0 * B * (128 * 2 * n + (1 if lean else 3) * n * n)
                print("dual softmax B=%d n=%d %-6s %-5s %.3f ms  (%.2f TB/s of algorithmic bytes)%s" % (
                    B, n, "split" if split else "exact", "lean" if lean else "full", t, nbytes / t / 1e9,
                    "  [pass 2: %s column chunks per row block]" % (chunks or "ntb / 2") if split else ""))
        ops.dual_softmax_set_chunks(0)
    if what in ("all", "sinkhorn"):
        B, n = 8, 4641
        d0 = torch.nn.functional.normalize(torch.randn((B, 128, n), generator=g), dim=1).to(dev)
        d1 = torch.nn.functional.normalize(torch.randn((B, 128, n), generator=g), dim=1).to(dev)
        s0 = (torch.rand((B, 1, n), generator=g) / n).to(dev)
        s1 = (torch.rand((B, 1, n), generator=g) / n).to(dev)
        ref = None
        for grp in (0, 1, 2, 4):
            ops.sinkhorn_set_group(grp)
            t = timed(lambda: ops.sinkhorn(d0, d1, 1.0, 10, s0, s1, want_scores=True, want_kp=True, want_final=True), reps=7)
            out = ops.sinkhorn(d0, d1, 1.0, 10, s0, s1, want_scores=True, want_kp=False, want_final=False)[0]
            if ref is None:
                ref = out
            print("sinkhorn B=%d n=%d group %2d: %.3f ms   bit-equal to the batch-wide order: %s" % (B, n, grp, t, bool(torch.equal(out, ref))))
        ops.sinkhorn_set_group(0)


if __name__ == "__main__":
    main()
