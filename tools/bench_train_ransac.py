"""Row N3 measurement: the training-time RANSAC loss (forward + backward) on the GPU drop-in, per stage, next to the CPU
oracle on a bounded sample.  Map-free training shape: n = 1938 keypoints per image (540 x 720 / 14), default LOSS_CLASS
constants (20 x 20 hypotheses per pair, 512 sampled matches, 8-point samples, 4 refinement steps).

    python tools/bench_train_ransac.py [--pairs 8] [--cpu-pairs 1]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops  # noqa: E402
from mickey_amd.config import _wrap  # noqa: E402
from mickey_amd.train_ransac import MetricPoseLoss  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--cpu-pairs", type=int, default=1)
    ap.add_argument("--n", type=int, default=1938)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    from oracle import train_oracle as TO   # CPU baseline leg + the synthetic batch generator
    cfg = TO.default_loss_cfg()
    batch = TO.synthetic_batch(args.pairs, args.n, seed=1, noise=0.12)
    dev = torch.device("cuda:0")
    b = {k: v.to(dev) for k, v in batch.items()}
    loss = MetricPoseLoss(_wrap(cfg), seed=1)

    def step():
        avg, outputs, grads, nvalid = loss(b)
        avg.backward()
        return avg, nvalid

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        avg, nvalid = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    # per-stage device time of the HIP entry points (HIP events on the current stream)
    stages = {}
    B, n = args.pairs, args.n
    rowp = b["final_scores"].reshape(B, n * n)

    def timed(name, fn, reps=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        stages[name] = e0.elapsed_time(e1) / reps
        return out

    idx, cnt = timed("sampler (mk_exprace_topk, 20 x 512 of n*n per pair)", lambda: ops.exprace_topk(rowp, 20, 512, seed=1))
    X, Y, w, _ = ops.gather_backproject(idx, b["final_scores"], b["kps0"], b["depth_kp0"], b["kps1"], b["depth_kp1"],
                                        b["K_color0"], b["K_color1"], 20)
    timed("hypotheses + refinement (mk_train_ransac_masks, %d hypotheses)" % (B * 400),
          lambda: ops.train_ransac_masks(X, Y, w, 20, 0.15, 4, 8, seed=1, offset=1))
    lv = torch.rand(B * 20, device=dev)
    timed("REINFORCE scatter (mk_reinforce_scatter)", lambda: ops.reinforce_scatter(idx, lv, B, 20, n * n))
    mask, _, _ = ops.train_ransac_masks(X, Y, w, 20, 0.15, 4, 8, seed=1, offset=1)
    T = b["T_0to1"].float()
    Rg, tg = T[:, :3, :3].reshape(B, 9).contiguous(), T[:, :3, 3].reshape(B, 3).contiguous()
    K0, K1 = b["Kori_color0"].float().reshape(B, 9).contiguous(), b["Kori_color1"].float().reshape(B, 9).contiguous()
    o, Rt, saved = timed("differentiable tail, forward (mk_train_tail_fwd: Procrustes + SVD + score + VCRE of %d hypotheses)" % (B * 400),
                         lambda: ops.train_tail_fwd(X, Y, mask, Rg, tg, K0, K1, 20, 20, 0.5, 0, True))
    gg = torch.rand((B * 400, 2), device=dev)
    timed("differentiable tail, backward (mk_train_tail_bwd)",
          lambda: ops.train_tail_bwd(X, Y, mask, Rg, tg, K0, K1, 20, 20, 0.5, 0, True, Rt, saved, gg))
    out = {"what": "MetricPoseLoss forward + backward (training-time RANSAC, SURVEY row N3)", "pairs": B, "keypoints": n,
           "ms_per_step": ms, "pairs_per_s": B / ms * 1e3, "avg_loss": float(avg.detach()), "valid": nvalid,
           "hip_stage_ms": {k: round(v, 3) for k, v in stages.items()},
           "remainder_ms (gather + back-projection and aggregation kernels, forward and backward; autograd and host launches)": round(ms - sum(stages.values()), 3)}
    if args.cpu_pairs > 0:
        cb = TO.synthetic_batch(args.cpu_pairs, args.n, seed=1, noise=0.12)
        g = torch.Generator().manual_seed(0)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            avg_c, *_ = TO.ransac_vectorized(cb, cfg, generator=g)
            avg_c.backward()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out["cpu_oracle"] = {"pairs": args.cpu_pairs, "s_per_step_median_of_3": ts[1], "pairs_per_s": args.cpu_pairs / ts[1],
                             "cores": torch.get_num_threads()}
        out["gpu_over_cpu"] = out["pairs_per_s"] / out["cpu_oracle"]["pairs_per_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
