"""MicKey hot-path throughput on MI355X:  python bench.py --gpus N --steps K --warmup W [--batch B]

One "step" = one full MickeyRelativePose.forward (ViT-L/14 encoder + 4 heads for both images,
dual-softmax matcher, 20x100-hypothesis probabilistic-Procrustes RANSAC) over a batch of B synthetic
540x720 (W x H) image pairs per GPU, inputs already resident in HBM, followed by the single RCCL
all-gather of the poses when N > 1.  Weak scaling: every rank processes its own B pairs.
Rank 0 prints ONE JSON line (contract in the task statement) with
  roofline     -- the dominant kernel (the 16-bit MFMA GEMM of the encoder linears): algorithmic FLOPs of
                  the launches in the timed region / their summed HIP-event durations, vs 2.5 PFLOP/s
  cpu_baseline -- the CPU oracle (torch-CPU fp32 restatement of the reference) timed on this box's host
                  cores on a bounded sample (1 pair), N = 1 only
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_TF = 2500.0   # dense bf16/fp16, MI355X_MICROARCH.md
H, W = 720, 540


def pmc_traffic_bytes(batch):
    """HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes of THIS workload
    (profiles/r01_pmc_traffic.json, produced by tools/pmc_bench_traffic.sh; FETCH_SIZE doubled as the gfx950 note in
    MI355X_MICROARCH.md prescribes for 16-B/lane streaming reads).  Counters cannot be collected from inside the
    timed run, so the figure is only reported for the batch size it was measured at."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if batch != 32 or not os.path.exists(path):
        return None
    d = json.load(open(path))
    try:
        return (2.0 * d["FETCH_SIZE"]["gemm_dense_256"]["mean_KiB"] + d["WRITE_SIZE"]["gemm_dense_256"]["mean_KiB"]) * 1024.0
    except KeyError:
        return None


class GemmProfiler:
    """HIP-event timing of every encoder-linear GEMM launch on the stream it is launched on
    (torch's current stream; ops.* launch there)."""

    def __init__(self):
        self.records = []   # (flops, ev0, ev1)
        self.on = False

    def wrap(self, ops):
        prof = self

        def timed(fn, flops_of):
            def inner(*a, **k):
                if not prof.on:
                    return fn(*a, **k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **k)
                e1.record()
                prof.records.append((flops_of(*a, **k), e0, e1))
                return out
            return inner
        ops.gemm = timed(ops.gemm, lambda a, w, *r, **k: 2.0 * a.shape[0] * w.shape[0] * w.shape[1])
        ops.gemm_ls_residual = timed(ops.gemm_ls_residual, lambda a, w, *r, **k: 2.0 * a.shape[0] * w.shape[0] * w.shape[1])
        ops.gemm_qkv = timed(ops.gemm_qkv, lambda a, w, *r, **k: 2.0 * a.shape[0] * w.shape[0] * w.shape[1])

    def summary(self):
        if not self.records:
            return None
        fl = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return {"launches": len(self.records), "flops": fl, "ms": ms, "tflops": fl / (ms * 1e-3) / 1e12,
                "avg_launch_ms": ms / len(self.records), "avg_launch_gflop": fl / len(self.records) / 1e9}


def cpu_baseline(cfg, sd):
    """The oracle timed on the host cores: 1 pair, full forward, fp32 (bounded sample)."""
    from mickey_amd import synthetic as syn
    from oracle import mickey_oracle as O
    cores = min(os.cpu_count() or 1, 32)   # torch-CPU does not scale past ~32 threads on these ops (256 -> 10x slower)
    torch.set_num_threads(cores)
    data = syn.synthetic_batch(B=1, H=H, W=W, seed=1234)
    t0 = time.time()
    with torch.no_grad():
        O.mickey_forward(sd, cfg, data)
    dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "1 pair 540x720, full forward (ViT-L fp32 + heads + dual-softmax + 20x100 RANSAC), torch-CPU "
                      "oracle, %.1f s" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="image pairs per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--gemm-tile", type=int, default=0, help="mk_gemm_set_tile mode (0 = automatic)")
    ap.add_argument("--attn-mode", type=int, default=0, help="mk_attn_set_mode mode (0 = default)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="hipGraph replay of the forward (auto: batches of <= 4 pairs, where launches dominate)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ   # launched by torch.distributed.run (any world size)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    from mickey_amd import distributed as D
    from mickey_amd import ops, synthetic as syn
    from mickey_amd.config import default_cfg
    from mickey_amd.model import MickeyRelativePose

    ops.gemm_set_tile(args.gemm_tile)
    ops.attn_set_mode(args.attn_mode)
    cfg = default_cfg()
    cfg["AMD"]["ENCODER_DTYPE"] = args.dtype
    cfg["AMD"]["SEED"] = rank
    cfg["AMD"]["GRAPH"] = {"auto": "auto", "on": True, "off": False}[args.graph]
    sd = syn.mickey_state_dict(cfg, seed=0)
    model = MickeyRelativePose(cfg)
    model.load_state_dict(sd)
    model = model.to(dev)
    B = args.batch
    batch = syn.synthetic_batch(B=B, H=H, W=W, seed=1234 + 2 * rank)
    data0 = {k: v.to(dev) for k, v in batch.items()}
    prof = GemmProfiler()
    if not args.no_kernel_events:
        prof.wrap(ops)

    def step():
        data = dict(data0)
        R, t = model(data)
        if use_dist:
            data["poses_all"] = D.gather_poses(R, t, data["inliers"])   # the one collective: [B_local,13] -> [world*B_local,13]
        return data

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    prof.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof.on = False
    graphed = len(model._graphs) > 0
    if graphed and not args.no_kernel_events:
        # a replayed graph bypasses the Python-level launch wrappers: time the GEMM launches of ONE extra eager step
        model.graph_mode = False
        prof.on = True
        step()
        torch.cuda.synchronize()
        prof.on = False
    if use_dist:
        assert last["poses_all"][0].shape[0] == world * B
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ok = bool(torch.isfinite(last["R"]).all())

    if rank == 0:
        g = prof.summary()
        roof = None
        if g:
            roof = {"bound": "mfma", "kernel": "gemm_pp64_kernel<%s> (encoder linears: qkv, proj, fc1, fc2)" % args.dtype,
                    "achieved": g["tflops"], "peak": PEAK_MFMA_TF, "unit": "TFLOP/s", "frac": g["tflops"] / PEAK_MFMA_TF,
                    "traffic": pmc_traffic_bytes(B), "traffic_unit": "bytes/launch (L2-miss side, PMC)",
                    "algorithmic_bytes_per_launch": 1.403e9 * B / 32.0, "launches": g["launches"], "avg_launch_ms": g["avg_launch_ms"],
                    "avg_launch_gflop": g["avg_launch_gflop"]}
            if graphed:
                roof["note"] = "forward replayed as a hipGraph in the timed region; kernel events from one extra eager step"
        out = {
            "metric": "image pairs/sec (540x720)", "value": world * B * args.steps / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "MickeyRelativePose.forward, ViT-L/14 + 4 heads + dual-softmax + 20x100-hypothesis "
                                   "Procrustes RANSAC, %d pairs/GPU of 540x720 (W x H), random-init weights" % B,
                       "pairs_per_gpu": B, "global_batch": world * B, "image_hw": [H, W], "keypoints": 1938,
                       "hypotheses": 2000, "parallelism": "pairs sharded over %d GPU(s), 1 all-gather of poses" % world,
                       "hip_graph": graphed},
            "roofline": roof,
            "finite_output": ok,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
