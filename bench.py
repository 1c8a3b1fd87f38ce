"""MicKey hot-path throughput on MI355X:  python bench.py --gpus N --steps K --warmup W [--batch B]

One "step" = one full MickeyRelativePose.forward (ViT-L/14 encoder + 4 heads for both images,
dual-softmax matcher, 20x100-hypothesis probabilistic-Procrustes RANSAC) over a batch of B synthetic
540x720 (W x H) image pairs per GPU, inputs already resident in HBM, followed by the single RCCL
all-gather of the poses when N > 1 (issued on a side stream, waited for before the step's results are
read).  Weak scaling: every rank processes its own B pairs.

``--gpus N`` with N > 1 launches N ranks itself (one process per GPU through torch.distributed.run,
rendezvous on 127.0.0.1) unless the script already runs under torch.distributed.run; it fails loudly when
the node has fewer than N GPUs.  ``n_gpus`` in the output is the number of ranks RCCL saw.

Rank 0 prints ONE COMPACT JSON line (< 3 KB: the driver keeps an 8-KB tail of stdout) as the LAST line of stdout with
  roofline     -- the dominant kernel (the 16-bit MFMA GEMM of the encoder linears): algorithmic FLOPs of
                  the launches in the timed region / their summed HIP-event durations, vs 2.5 PFLOP/s; `traffic` = HBM-side
                  bytes per launch from the committed PMC passes of the same kernel sources (profiles/pmc_traffic.json: a
                  per-source-hash constant); `peak_sustained` / `frac_sustained` = what the MFMA instruction alone
                  sustained on this box straight after the timed region (mk_dev_mfma_sustained; information, never `peak`)
  stages       -- the six largest stages: ms per step and fraction of their governing peak
  cpu_baseline -- the CPU oracle (torch-CPU fp32 restatement of the reference) timed on this box's host
                  cores on a bounded sample (1 pair; 1 warm-up, then the median of 3 runs; per-stage seconds), N = 1 only
  sustained_60 -- the headline configuration over 60 back-to-back steps straight after the timed region, pairs/s
  value_ref_precision -- the same forward at the reference's literal precision split (fp16 ViT + fp32-grade heads,
                  mickey_extractor.py:49-56; leg `ref_split`), pairs/s
  natural_operands -- the same forward on 1/f-spectrum images and massive-activation weights (leg `natural`): pairs/s and the
                  encoder GEMM's TFLOP/s; information about the power limit, never `value`
  single_pair_ms -- BASELINE.json configs[1]: one 540x720 pair, hipGraph replay
Everything else -- the full per-stage roofline list, every leg (--legs all: fp16 everywhere, ref_split, ref_split_fp32mfma,
attn_mfma16, vit_small, config5, natural), --include-h2d (PCIe-inclusive), --precision (errors vs the oracle
outputs of the same run) -- goes to gpurun_out/bench_detail.json (--detail PATH) and is summarised on stderr.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_TF = 2500.0   # dense bf16/fp16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0   # HBM3E spec, MI355X_MICROARCH.md
H, W = 720, 540
TRAFFIC_SOURCE = "profiles/pmc_traffic.json"   # written by tools/profile_round.sh; keyed by the kernel-source hash


STAGE_ROLES = {"encoder_gemm": ("encoder_gemm",), "attention": ("attention",), "conv_gemm": ("conv_gemm",),
               "matcher": ("matcher_",), "sampler": ("sampler",), "layernorm": ("layernorm",)}


def pmc_stage_traffic(batch):
    """{stage: HBM-side bytes per forward} from the same committed counter file (per kernel role, summed over the launches of
    one forward), under the same conditions as pmc_traffic(): same kernel sources, same batch; else {}."""
    from mickey_amd import build as mkbuild
    path = os.path.join(ROOT, TRAFFIC_SOURCE)
    if not os.path.exists(path):
        return {}
    d = json.load(open(path))
    if d.get("source_hash") != mkbuild.source_hash() or d.get("batch") != batch:
        return {}
    out = {}
    for stage, prefixes in STAGE_ROLES.items():
        vals = [v["bytes_per_forward"] for k, v in d.get("roles", {}).items()
                if any(k.startswith(p) for p in prefixes) and v.get("bytes_per_forward") is not None]
        if vals:
            out[stage] = sum(vals)
    return out


def pmc_traffic(batch):
    """HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes of THIS workload
    (TRAFFIC_SOURCE, produced by tools/profile_round.sh; FETCH_SIZE doubled as the gfx950 note in MI355X_MICROARCH.md
    prescribes for 16-B/lane streaming reads).  Counters cannot be collected from inside the timed run, so the figure is
    read from that file -- and only reported when the file was taken from the SAME kernel sources (build.source_hash())
    and batch size as this run; otherwise traffic is null."""
    from mickey_amd import build as mkbuild
    path = os.path.join(ROOT, TRAFFIC_SOURCE)
    if not os.path.exists(path):
        return None, "no %s" % TRAFFIC_SOURCE
    d = json.load(open(path))
    if d.get("source_hash") != mkbuild.source_hash():
        return None, "%s was collected from other kernel sources (%s != %s)" % (TRAFFIC_SOURCE, d.get("source_hash"), mkbuild.source_hash())
    if d.get("batch") != batch:
        return None, "%s was collected at batch %s" % (TRAFFIC_SOURCE, d.get("batch"))
    try:
        g = d["encoder_gemm"]
        return (2.0 * g["FETCH_SIZE_KiB_per_launch"] + g["WRITE_SIZE_KiB_per_launch"]) * 1024.0, None
    except KeyError as e:
        return None, "%s lacks %s" % (TRAFFIC_SOURCE, e)


class StageProfiler:
    """HIP-event timing of every hot-path launch on the stream it is launched on (torch's current stream;
    ops.* launch there), grouped into stages with their ALGORITHMIC work."""

    # stage -> (governing bound, unit of work)
    STAGES = {
        "encoder_gemm": ("mfma", "flop"), "attention": ("mfma", "flop"), "conv_gemm": ("mfma", "flop"),
        "head_gemm": ("hbm", "byte"), "split_planes": ("hbm", "byte"), "layernorm": ("hbm", "byte"), "matcher": ("hbm", "byte"),
        "sampler": ("hbm", "byte"),
        "hypotheses_refine": (None, None),
    }

    def __init__(self):
        self.records = []   # (stage, work, ev0, ev1)
        self.on = False

    def wrap(self, ops):
        import torch
        prof = self

        def timed(name, stage, work_of, bytes_of=None):
            fn = getattr(ops, name)

            def inner(*a, **k):
                if not prof.on:
                    return fn(*a, **k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **k)
                e1.record()
                prof.records.append((stage, float(work_of(*a, **k)), e0, e1, float(bytes_of(*a, **k)) if bytes_of else 0.0))
                return out
            setattr(ops, name, inner)

        esz = lambda t: t.element_size()  # noqa: E731
        mnk = lambda a, w, *r, **k: 2.0 * a.shape[0] * w.shape[0] * w.shape[1]  # noqa: E731
        # ALGORITHMIC bytes of one encoder-GEMM launch, from the shapes and plane widths actually passed: A and W read once,
        # the output written once, plus what the fused epilogue moves (fp32 residual read-modify-write, or the two 16-bit
        # planes of the split stream read + written, and the per-slot row statistics block written / read)
        opnd = lambda a, w: a.shape[0] * w.shape[1] * esz(a) + w.shape[0] * w.shape[1] * esz(w)  # noqa: E731
        stat = lambda M, D: M * (D // 64) * 8  # noqa: E731

        def b_gemm(a, w, bias=None, act=0, out_f32=False, out=None, **k):
            return opnd(a, w) + a.shape[0] * w.shape[0] * (esz(out) if out is not None else (4 if out_f32 else esz(a)))
        timed("gemm", "encoder_gemm", mnk, b_gemm)
        timed("gemm_ls_residual", "encoder_gemm", mnk, lambda a, w, bias, gamma, x: opnd(a, w) + 8 * x.numel())
        timed("gemm_qkv", "encoder_gemm", mnk, lambda a, w, *r, **k: opnd(a, w) + a.shape[0] * w.shape[0] * esz(a))
        timed("gemm_patch_embed", "encoder_gemm", mnk,
              lambda a, w, bias, pos, x, nimg, npatch: opnd(a, w) + a.shape[0] * w.shape[0] * 4 + pos.numel() * 4)
        # LayerNorm folded in
        timed("gemm_ln", "encoder_gemm", mnk, lambda a, w, bias, colsum, stats, eps, act=0, out=None, **k:
              opnd(a, w) + a.shape[0] * w.shape[0] * esz(a) + stat(a.shape[0], w.shape[1]))
        timed("gemm_qkv_ln", "encoder_gemm", mnk, lambda a, w, bias, colsum, stats, *r, **k:
              opnd(a, w) + a.shape[0] * w.shape[0] * esz(a) + stat(a.shape[0], w.shape[1]))
        timed("gemm_ls_residual_ln", "encoder_gemm", mnk, lambda a, w, bias, gamma, xh, xl, stats, x_out=None, **k:
              opnd(a, w) + 2 * esz(xh) * xh.numel() + (4 * xh.numel() if x_out is not None else
                                                      2 * esz(xh) * xh.numel() + stat(a.shape[0], w.shape[0])))
        timed("gemm_patch_embed_ln", "encoder_gemm", mnk, lambda a, w, bias, pos, xh, xl, stats, nimg, npatch:
              opnd(a, w) + a.shape[0] * w.shape[0] * 2 * esz(xh) + pos.numel() * 4 + stat(a.shape[0], w.shape[0]))
        # attention.py:53-59: per (image, head) QK^T and PV, 2 * 2 * N^2 * 64
        timed("flash_attn", "attention", lambda q, k, vt, out, nimg, heads, ntok, pad: 4.0 * nimg * heads * ntok * ntok * 64)
        # implicit-GEMM 3x3 conv: 2 * M * Cout * (9 C1 + C2) per group
        timed("conv3x3", "conv_gemm",
              lambda in1, C1, w, bias, out, Cout, groups, nimg, Hh, Ww, act=0, in2=None, C2=0, **k:
              2.0 * groups * nimg * Hh * Ww * Cout * (9 * C1 + (C2 if in2 is not None else 0)))
        # split-operand convs (AMD.HEADS_DTYPE: split): the SAME algorithmic flops -- the three MFMA products per multiply are the
        # implementation's -- and the fp32 -> (hi, lo) plane conversions in front of them
        timed("conv3x3_split", "conv_gemm",
              lambda in1, C1, w, bias, out, Cout, groups, nimg, Hh, Ww, act=0, in2=None, C2=0, **k:
              2.0 * groups * nimg * Hh * Ww * Cout * (9 * C1 + (C2 if in2 is not None else 0)))
        timed("split_planes", "split_planes", lambda x, hi, lo, scale=64.0, **k: 8.0 * x.numel())
        timed("gemm_grouped_split", "head_gemm", lambda a, w, bias, out, groups, M, N, K, *r, **k:
              float(groups) * (M * K * 4 + N * 2 * K * 2 + M * N * 4))
        # the head linears (K = 128..256, fp32 qkv outputs) are write-bound: 1 flop per 3 bytes; priced against HBM
        timed("gemm_grouped", "head_gemm", lambda a, w, bias, out, groups, M, N, K, *r, **k:
              float(groups) * (M * K * esz(a) + N * K * esz(w) + M * N * esz(out)))

        # Linear(-> 128) + LayerNorm (+ residual) in one pass: A and W read, the 16-bit rows written, the fp32 residual read + written
        timed("gemm_ln128", "head_gemm", lambda a, w, ln_w, ln_b, eps, out, groups, M, K, lda=None, ldo=None, resid=None, bordered=None:
              float(groups) * (M * K * esz(a) + 128 * K * esz(w) + M * 128 * (esz(out) + (8 if resid is not None else 0))))

        def ln_bytes(x, w, b, eps, out=None, out_dtype=None, resid=None, rows_out=None, **k):
            D = w.shape[-1]
            rows = rows_out if rows_out is not None else x.numel() // x.shape[-1]
            ob = 4 if isinstance(out, (tuple, list)) else (esz(out) if out is not None else 2)   # (hi, lo) fp16 planes: 2 + 2 bytes
            return rows * D * (4 + ob + (4 if resid is not None else 0))
        timed("layernorm", "layernorm", ln_bytes)

        def match_bytes(dsc0, dsc1, scr0=None, scr1=None, temperature=0.1, dustbin=None, want_scores=True, want_kp=True,
                        want_final=True, split=False):
            B, C, n0 = dsc0.shape
            n1 = dsc1.shape[2]
            nout = int(want_scores) + int(want_kp and scr0 is not None) + int(want_final and scr0 is not None)
            return 4.0 * B * (C * (n0 + n1) + nout * n0 * n1)
        timed("dual_softmax", "matcher", match_bytes)
        timed("sinkhorn", "matcher", lambda dsc0, dsc1, alpha, iters=10, *r, **k:
              4.0 * dsc0.shape[0] * (2 * iters + 1) * (dsc0.shape[2] + 1) * (dsc1.shape[2] + 1))
        timed("exprace_topk", "sampler", lambda p, rows, kk, **k: 4.0 * p.numel() + 4.0 * p.shape[0] * rows * kk)
        for nm in ("gather_backproject", "ransac_hypotheses", "refine_pose"):
            timed(nm, "hypotheses_refine", lambda *a, **k: 0.0)

    def summary(self, steps):
        by = {}
        for stage, work, e0, e1, nbytes in self.records:
            d = by.setdefault(stage, {"launches": 0, "work": 0.0, "ms": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["work"] += work
            d["bytes"] += nbytes
            d["ms"] += e0.elapsed_time(e1)
        total_ms = sum(d["ms"] for d in by.values()) or 1.0
        out = []
        for stage, (bound, unit) in self.STAGES.items():
            d = by.get(stage)
            if not d:
                continue
            ent = {"stage": stage, "bound": bound, "launches_per_step": d["launches"] / steps, "ms_per_step": d["ms"] / steps,
                   "share_of_kernel_time": d["ms"] / total_ms}
            if bound == "mfma":
                ach = d["work"] / (d["ms"] * 1e-3) / 1e12
                ent.update(achieved=ach, peak=PEAK_MFMA_TF, unit="TFLOP/s", frac=ach / PEAK_MFMA_TF,
                           algorithmic_gflop_per_step=d["work"] / steps / 1e9)
                if d["bytes"] > 0:   # the same launches against the OTHER roof (narrow models: K = 384 linears are as much HBM as MFMA)
                    ent.update(algorithmic_mb_per_step=d["bytes"] / steps / 1e6,
                               hbm_frac=d["bytes"] / (d["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS)
            elif bound == "hbm":
                ach = d["work"] / (d["ms"] * 1e-3) / 1e9
                ent.update(achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                           algorithmic_mb_per_step=d["work"] / steps / 1e6)
            else:
                ent.update(achieved=None, peak=None, unit=None, frac=None, note="latency / VALU bound, LDS-resident: no roofline claim")
            out.append(ent)
        return out, by


def cpu_baseline(cfg, sd):
    """The oracle timed on the host cores: 1 pair, full forward, fp32; one warm-up,
    then the median of 3 runs, per stage (encoder / heads / matcher / solver) and in total."""
    import torch
    from mickey_amd import synthetic as syn
    from oracle import mickey_oracle as O
    cores = min(os.cpu_count() or 1, 32)   # torch-CPU does not scale past ~32 threads on these ops (256 -> 10x slower)
    torch.set_num_threads(cores)
    pre = "compute_matches.extractor."
    f = cfg["MICKEY"]["DINOV2"]["DOWN_FACTOR"]

    def one(timed):
        data = syn.synthetic_batch(B=1, H=H, W=W, seed=1234)
        t = {}
        with torch.no_grad():
            t0 = time.perf_counter()
            feats = []
            for key in ("image0", "image1"):
                img = data[key][:, :, : f * (H // f), : f * (W // f)]
                tok = O.vit_forward_features(sd, pre + "dinov2_vitl14.", img, 16)
                feats.append(tok.permute(0, 2, 1).reshape(1, tok.shape[-1], H // f, W // f).float())
            t1 = time.perf_counter()
            per = [O.extractor_heads(sd, cfg, ft, pre) for ft in feats]
            t2 = time.perf_counter()
            out = {}
            for i, (kp, dp, sc, ds) in enumerate(per):
                kp = O.abs_keypoints(kp, f)
                n = kp.shape[2] * kp.shape[3]
                out["kps%d" % i], out["depth_kp%d" % i] = kp.reshape(1, 2, n), dp.reshape(1, 1, n)
                out["scr%d" % i], out["dsc%d" % i] = sc.reshape(1, 1, n), ds.reshape(1, ds.shape[1], n)
            dust = sd.get("compute_matches.matcher.matching_mat.dustbin_score")
            out["scores"] = O.dual_softmax(out["dsc0"], out["dsc1"], dust, cfg["FEATURE_MATCHER"]["DUAL_SOFTMAX"]["TEMPERATURE"])
            out["final_scores"] = out["scores"] * torch.matmul(out["scr0"].transpose(2, 1).contiguous(), out["scr1"])
            t3 = time.perf_counter()
            data.update(out)
            O.estimate_pose(data, cfg)
            t4 = time.perf_counter()
            cpu_baseline.last_outputs = {k: v for k, v in out.items()}   # the oracle's features of this pair (precision report)
        t.update(encoder=t1 - t0, heads=t2 - t1, matcher=t3 - t2, solver=t4 - t3, total=t4 - t0)
        return t

    warm = one(False)   # warm-up (thread pools, allocator, first-touch of the weights)
    runs = sorted([one(True) for _ in range(3)], key=lambda r: r["total"])
    med = runs[1]       # the MEDIAN of 3 runs behind one warm-up (~7 s each on 32 cores: ~28 s of CPU work)
    # thread-count probe (opt-in, --cpu-threads-probe): the same oracle on a SMALL pair (182x196) at the thread count used above and
    # at os.cpu_count().  Not in the default run: measured once on the 256-core GPU box (profiles/r06g_bench_detail.json), the
    # 182x196 pair took 0.61 s on 32 threads and 250 s on 256 -- torch-CPU collapses when every core takes part in these ops
    probe = {"measured_once": {"box_cores": 256, "sample": "1 pair 182x196, same oracle", "seconds_threads_32": 0.6115,
                               "seconds_threads_256": 249.97, "source": "profiles/r06g_bench_detail.json"}}
    if getattr(cpu_baseline, "threads_probe", False):
        try:
            ncpu = os.cpu_count() or 1
            small = syn.synthetic_batch(B=1, H=182, W=196, seed=1234)

            def small_forward():
                d = {k: v.clone() for k, v in small.items()}
                with torch.no_grad():
                    t0 = time.perf_counter()
                    d.update(O.compute_correspondences(sd, cfg, d))
                    O.estimate_pose(d, cfg)
                    return time.perf_counter() - t0
            probe["sample"] = "1 pair 182x196, same oracle, 1 warm-up + 1 timed run per setting"
            for name, nt in (("threads_%d" % cores, cores), ("threads_%d" % ncpu, ncpu)):
                if ("seconds_" + name) in probe:
                    continue
                torch.set_num_threads(nt)
                small_forward()
                probe["seconds_" + name] = round(small_forward(), 4)
            torch.set_num_threads(cores)
        except Exception as e:   # the probe never costs the baseline itself
            probe["error"] = "%s: %s" % (type(e).__name__, e)
    return {"value": 1.0 / med["total"], "unit": "pairs/s", "cores": cores, "cores_available": os.cpu_count(), "kind": "port",
            "kind_note": "the oracle restatement, not the reference module itself: /root/reference does not exist on the GPU box; "
                         "threads capped at 32 because torch-CPU collapses beyond that on these ops (threads_probe: a 182x196 pair 0.61 s on 32 "
                         "threads, 250 s on all 256).  The only "
                         "reference-side measurement is the survey's: the reference module itself, 0.055 pairs/s on the 8 cores "
                         "of the build container (BASELINE.md section 2)",
            "pinned_by": "tests/test_oracle_golden.py (the oracle vs the reference's own outputs, tests/golden/*.npz, regenerated "
                         "from /root/reference by oracle/make_golden.py in test_committed_fixtures_reproduce_from_the_reference)",
            "protocol": "1 warm-up + 3 timed runs, the median counts", "stage_seconds": {k: round(v, 4) for k, v in med.items()},
            "runs_total_seconds": [round(r["total"], 4) for r in runs], "threads_probe": probe,
            "sample": "1 pair 540x720, full forward (ViT-L fp32 + heads + dual-softmax + 20x100 RANSAC), torch-CPU "
                      "oracle, %.2f s per pair (median of 3 after 1 warm-up)" % med["total"]}


def precision_report(make_model, syn, dev, args, oracle_out):
    """rel-Frobenius error of every feature output vs the CPU oracle (fp32) on the one 540x720 pair the cpu_baseline leg
    computed in this run, for the headline configuration and the fp16 leg, next to the committed oracle-side noise floors
    (tests/golden/noise_floor_lp.npz, oracle/make_noise_floor.py)."""
    import numpy as np
    import torch
    keys = ("kps0", "depth_kp0", "scr0", "dsc0", "scores", "final_scores")
    floors = None
    fp = os.path.join(ROOT, "tests", "golden", "noise_floor_lp.npz")
    if os.path.exists(fp):
        floors = np.load(fp)
    rep = {}
    for name, dtype, floor_key in (("headline", args.dtype, "bf16_encheads_720_" if args.dtype == "bf16" else "ref_fp16_720_"),
                                   ("fp16", "fp16", "ref_fp16_720_")):
        if name == "fp16" and args.dtype == "fp16":
            rep["fp16"] = rep["headline"]
            continue
        m = make_model(dtype)[0]
        d = {k: v.to(dev) for k, v in syn.synthetic_batch(B=1, H=H, W=W, seed=1234).items()}
        m.compute_correspondences(d)
        torch.cuda.synchronize()
        err = {}
        for k in keys:
            a, b = d[k].double().cpu(), oracle_out[k].double()
            err[k] = float((a - b).norm() / (b.norm() + 1e-30))
        ent = {"dtype": dtype, "heads_operands": str(m.heads_dtype).replace("torch.", ""), "error_vs_oracle": err}
        if floors is not None:
            ent["floor"] = {k: float(floors[floor_key + k]) for k in keys}
            ent["floor_source"] = "tests/golden/noise_floor_lp.npz:" + floor_key + "*"
            ent["inside_floor"] = all(err[k] <= ent["floor"][k] for k in keys)
        rep[name] = ent
        del m, d
        torch.cuda.empty_cache()
    return rep


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """Re-execute this script as n ranks (one per GPU) under torch.distributed.run on 127.0.0.1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="image pairs per GPU per step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--heads-dtype", default="auto", choices=["auto", "same", "bf16", "fp16", "fp32", "split"],
                    help="AMD.HEADS_DTYPE of the headline model: operand type of the four head stacks (auto = fp16 beside a "
                         "16-bit encoder; the reference runs them in fp32)")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin the ranks to their GPUs' NUMA cores")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads-probe", action="store_true", help="CPU leg: also time a small pair at os.cpu_count() threads "
                                                                     "(minutes on a 256-core host)")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--legs", default="ref_split,natural",
                    help="comma-separated extra legs, each a full timed run written to the detail file: fp16, ref_split, "
                         "ref_split_fp32mfma, attn_mfma16, vit_small, config5, natural; 'all'; 'none'.  Default: ref_split (the "
                         "reference's literal precision split, reported in the line as value_ref_precision) and natural (1/f "
                         "images + massive-activation weights: natural_operands of the line)")
    ap.add_argument("--sustained", action="store_true", help="(default since round 6; kept for old command lines)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the 60-step run of the headline configuration ('sustained_60' of the line)")
    ap.add_argument("--precision", action="store_true",
                    help="also report every feature output's error vs the CPU oracle outputs of this run (detail file)")
    ap.add_argument("--lean", action="store_true", help="only the headline measurement: no legs, no single-pair / CPU legs "
                                                         "(profiling passes)")
    ap.add_argument("--include-h2d", action="store_true",
                    help="also report the PCIe-inclusive rate under 'pcie_inclusive' (detail file): uint8 frames from host "
                         "memory through the input pipeline (pinned ring, H2D, resize kernel) into the forward; never `value`")
    ap.add_argument("--no-single", action="store_true", help="skip the one-pair latency leg reported as 'single_pair_ms'")
    ap.add_argument("--detail", default=None, help="path of the detail JSON (default gpurun_out/bench_detail.json; 'none' = do not write)")
    ap.add_argument("--no-ln-fold", action="store_true", help="dev: stand-alone LayerNorm kernels instead of the folded form (A/B)")
    ap.add_argument("--no-half-rows", action="store_true", help="dev: never pick the 64x128 GEMM tiling automatically (A/B)")
    ap.add_argument("--gemm-tile", type=int, default=0, help="dev: mk_gemm_set_tile mode (0 = automatic)")
    ap.add_argument("--attn-mode", type=int, default=0, help="dev: mk_attn_set_mode mode (0 = default)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="hipGraph replay of the forward (auto: batches of <= 4 pairs, where launches dominate)")
    ap.add_argument("--dump-poses", default=None, help="TEST HOOK (tests/test_rccl_gpu.py): torch.save the poses of the last "
                                                        "timed step (the gathered ones under torch.distributed.run) to this path")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend ('nccl' is RCCL on ROCm)")
    ap.add_argument("--stub", action="store_true",
                    help="TEST HOOK (tests/test_bench_cpu.py): CPU tensors and a trivial stand-in model, so that the launcher, "
                         "sharding, gather and timing logic of --gpus N can be exercised under gloo without a GPU; "
                         "the printed line is marked \"stub\": true and is not a measurement")
    args = ap.parse_args(argv)
    args.sustained = not args.no_sustained
    if args.lean:
        args.no_single = args.no_cpu_baseline = True
        args.include_h2d = args.sustained = args.precision = False
        args.legs = "none"
    all_legs = ("fp16", "ref_split", "ref_split_fp32mfma", "attn_mfma16", "vit_small", "config5", "natural")
    args.leg_set = set(all_legs) if args.legs == "all" else set() if args.legs in ("none", "") else set(args.legs.split(","))
    unknown = args.leg_set - set(all_legs)
    if unknown:
        ap.error("unknown leg(s) %s; known: %s" % (sorted(unknown), ", ".join(all_legs)))
    return args


def resolve_world(args, env, device_count):
    """-> (mode, world) with mode in {'single', 'spawn', 'rank'}; raises SystemExit with a clear message when the
    request cannot be honoured.  Pure function of its inputs (unit-tested on CPU)."""
    under_launcher = "RANK" in env and "WORLD_SIZE" in env
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if under_launcher:
        world = int(env["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d: they must agree" % (world, args.gpus))
        if device_count < world:
            raise SystemExit("bench.py: %d ranks requested but only %d GPU(s) visible on this node" % (world, device_count))
        return "rank", world
    if args.gpus == 1:
        if device_count < 1:
            raise SystemExit("bench.py: no GPU visible")
        return "single", 1
    if device_count < args.gpus:
        raise SystemExit("bench.py: --gpus %d requested but only %d GPU(s) visible on this node" % (args.gpus, device_count))
    return "spawn", args.gpus


class _StubModel:
    """--stub only: pose = deterministic function of the pair's image mean (no GPU, no kernels)."""

    _graphs = {}

    def __call__(self, data):
        import torch
        B = data["image0"].shape[0]
        m = data["image0"].reshape(B, -1).mean(1)
        data["R"] = torch.eye(3).repeat(B, 1, 1) * m.view(B, 1, 1)
        data["t"] = torch.stack([m, 2 * m, 3 * m], 1).view(B, 1, 3)
        data["inliers"] = (10 * m).view(B, 1)
        return data["R"], data["t"]


def measure(model, data0, args, use_dist, world, gatherer, prof=None):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + synchronize; MAX over ranks."""
    import torch
    import torch.distributed as dist
    on_gpu = data0["image0"].is_cuda

    class _Sync:   # torch.cuda.synchronize on the GPU; nothing to wait for on CPU tensors (--stub)
        @staticmethod
        def synchronize():
            if on_gpu:
                torch.cuda.synchronize()

    def step():
        data = dict(data0)
        R, t = model(data)
        if gatherer is not None:
            data["_gather"] = gatherer.submit(R, t, data["inliers"])   # side stream; the main stream runs on
        return data

    last = None
    for _ in range(args.warmup):
        last = step()
        if gatherer is not None:
            gatherer.wait(last["_gather"])
    _Sync.synchronize()
    if use_dist:
        dist.barrier()
    _Sync.synchronize()
    if prof is not None:
        prof.on = True
    t0 = time.perf_counter()
    pending = []
    for _ in range(args.steps):
        last = step()
        if gatherer is not None:
            pending.append(last["_gather"])
    poses = None
    for h in pending:   # results are only needed here: every gather has had a whole forward to complete
        poses = gatherer.wait(h)
    _Sync.synchronize()
    if use_dist:
        dist.barrier()
    _Sync.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        prof.on = False
    if use_dist:
        tt = torch.tensor([dt], device=data0["image0"].device, dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)           # every rank's own wall time: the line reports min / max, `value` uses the MAX
        measure.per_rank_s = [float(t.item()) for t in every]
        dt = max(measure.per_rank_s)
    else:
        measure.per_rank_s = [dt]
    return dt, last, poses


def per_rank_ms(steps):
    """min / max / all of the ranks' own ms per step of the last measure() (the first thing to look at when a scaling run
    disappoints: one slow rank, or all of them)."""
    ms = [t / steps * 1e3 for t in getattr(measure, "per_rank_s", [])]
    return {"min": min(ms), "max": max(ms), "all": [round(m, 3) for m in ms]} if ms else None


def pin_this_rank(args, rank, world, use_dist):
    """N > 1, BEFORE init_process_group (the RCCL proxy threads inherit the mask): pin this rank to its GPU's NUMA cores (or
    an even split of the allowed cores), mickey_amd.distributed.pin_rank.  -> this rank's placement or None."""
    from mickey_amd import distributed as D
    if not use_dist or args.no_pin:
        return None
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    return D.pin_rank(int(os.environ.get("LOCAL_RANK", rank)), local_world)


def gather_placements(mine, rank, world):
    """After init_process_group: the list of all ranks' placements on rank 0 (None elsewhere / when nothing was pinned)."""
    import torch.distributed as dist
    if mine is None:
        return None
    every = [None] * world
    dist.all_gather_object(every, mine)
    return every if rank == 0 else None


def main_stub(args, rank, world, use_dist):
    """--stub: the same launcher / sharding / gather / timing code path on CPU tensors with a stand-in model."""
    import torch
    import torch.distributed as dist
    from mickey_amd import distributed as D
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    mine = pin_this_rank(args, rank, world, use_dist)
    if use_dist:
        dist.init_process_group(args.backend if args.backend != "nccl" else "gloo", rank=rank, world_size=world)
        world = dist.get_world_size()
    affinity = gather_placements(mine, rank, world)
    B = args.batch
    g = torch.Generator().manual_seed(1234 + 2 * rank)
    data0 = {"image0": torch.rand((B, 3, 8, 8), generator=g), "image1": torch.rand((B, 3, 8, 8), generator=g)}
    gatherer = D.PoseGatherer(None) if use_dist else None
    dt, last, poses = measure(_StubModel(), data0, args, use_dist, world, gatherer, None)
    if use_dist:
        assert poses is not None and poses[0].shape[0] == world * B, "gathered poses do not cover the global batch"
        # every rank's own poses sit at its slot of the gathered batch
        assert torch.equal(poses[0][rank * B:(rank + 1) * B], last["R"])
    if rank == 0:
        print(json.dumps({"metric": "image pairs/sec (540x720)", "stub": True, "value": world * B * args.steps / dt,
                          "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "scaling": "weak",
                          "per_rank_ms_per_step": per_rank_ms(args.steps),
                          "config": {"pairs_per_gpu": B, "global_batch": world * B, "affinity": affinity}},
                         separators=(",", ":")), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def roofline_entry(stages, by, B, dtype, model):
    """The `roofline` object of the JSON line: the dominant kernel (the encoder GEMM) + the per-stage list."""
    g = by.get("encoder_gemm")
    if not g:
        return None
    tf = g["work"] / (g["ms"] * 1e-3) / 1e12
    traffic, why = pmc_traffic(B)
    st_traffic = pmc_stage_traffic(B)
    for st in stages:   # per stage: HBM-side bytes per step from the PMC passes next to the algorithmic bytes (bandwidth-bound stages)
        if st["stage"] in st_traffic:
            st["traffic_mb_per_step"] = st_traffic[st["stage"]] / 1e6
            if st.get("algorithmic_mb_per_step"):
                st["traffic_over_algorithmic"] = st["traffic_mb_per_step"] / st["algorithmic_mb_per_step"]
    roof = {"bound": "mfma", "kernel": "gemm_pp64_kernel<%s> (encoder linears: qkv, proj, fc1, fc2, patch embed)" % dtype,
            "achieved": tf, "peak": PEAK_MFMA_TF, "unit": "TFLOP/s", "frac": tf / PEAK_MFMA_TF,
            "traffic": traffic, "traffic_unit": "bytes/launch (L2-miss side, PMC)",
            "traffic_source": (TRAFFIC_SOURCE + " (committed rocprofv3 --pmc passes of this workload at the same kernel "
                               "sources; a constant, not measured in this run)") if traffic is not None else why,
            "algorithmic_bytes_per_launch": g["bytes"] / g["launches"],
            "algorithmic_bytes_note": "from the shapes and plane widths of the launches of the timed region (A, W, outputs, "
                                      "residual planes, row-statistics blocks), averaged over the launch mix",
            "launches": g["launches"], "avg_launch_ms": g["ms"] / g["launches"],
            "avg_launch_gflop": g["work"] / g["launches"] / 1e9, "stages": stages}
    if getattr(model, "ln_fold", False):
        import torch
        if model.lp_dtype != torch.float32:
            roof["note"] = ("the encoder GEMM epilogues carry the folded LayerNorm (statistics, normalisation, split residual "
                            "stream): 48 LayerNorm passes per forward are gone from the 'layernorm' stage and their remaining "
                            "cost is inside this time (--no-ln-fold for the A/B)")
    return roof


_T0 = time.perf_counter()


def tick(what):
    """Wall-clock account of the run on stderr (what the driver's clock around the whole command is spent on)."""
    sys.stderr.write("bench.py t+%6.1fs %s\n" % (time.perf_counter() - _T0, what))
    sys.stderr.flush()


def main(argv=None):
    args = parse_args(argv)
    import torch
    tick("torch imported")
    mode, world = resolve_world(args, os.environ, args.gpus if args.stub else torch.cuda.device_count())
    if mode == "spawn":
        sys.exit(spawn_ranks(world, sys.argv[1:] if argv is None else list(argv)))

    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = mode == "rank"
    if args.stub:
        return main_stub(args, rank, world, use_dist)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    mine = pin_this_rank(args, rank, world, use_dist)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world, device_id=dev)
        world = dist.get_world_size()   # what RCCL actually saw
    affinity = gather_placements(mine, rank, world)

    from mickey_amd import distributed as D
    from mickey_amd import ops, synthetic as syn
    from mickey_amd.config import default_cfg
    from mickey_amd.model import MickeyRelativePose

    ops.gemm_set_tile(args.gemm_tile)
    if args.no_half_rows:
        ops.gemm_set_tile(500)
    ops.attn_set_mode(args.attn_mode)

    sd_cache = {}   # the synthetic checkpoint of an architecture is drawn once (350 M random numbers) and shared by the legs

    def make_model(dtype, heads=None, arch="vit_large", matcher=None, outliers=False):
        cfg = default_cfg()
        cfg["AMD"]["ENCODER_DTYPE"] = dtype
        cfg["AMD"]["HEADS_DTYPE"] = heads or args.heads_dtype
        cfg["AMD"]["SEED"] = rank
        cfg["AMD"]["LN_FOLD"] = not args.no_ln_fold
        cfg["AMD"]["GRAPH"] = {"auto": "auto", "on": True, "off": False}[args.graph]
        cfg["AMD"]["VIT"] = arch
        cfg["MICKEY"]["DINOV2"]["CHANNEL_DIM"] = syn.VIT_ARCH[arch][0]
        if matcher:
            cfg["FEATURE_MATCHER"]["TYPE"] = matcher
        key = (arch, cfg["FEATURE_MATCHER"]["TYPE"], bool(outliers))
        if key not in sd_cache:
            plain = sd_cache.get(key[:2] + (False,))
            if outliers and plain is not None:   # the same draw as mickey_state_dict(outliers=True): a copy with the outliers planted
                sd_cache[key] = {k: v.clone() for k, v in plain.items()}
                syn.plant_outliers(sd_cache[key], arch, syn.DINO_PREFIX, 0)
            else:
                sd_cache[key] = syn.mickey_state_dict(cfg, seed=0, arch=arch, outliers=bool(outliers))
        sd = sd_cache[key]
        m = MickeyRelativePose(cfg)
        m.load_state_dict(sd)
        return m.to(dev), cfg, sd

    prof = None
    if not args.no_kernel_events:
        prof = StageProfiler()
        prof.wrap(ops)

    model, cfg, sd = make_model(args.dtype)
    B = args.batch
    batch = syn.synthetic_batch(B=B, H=H, W=W, seed=1234 + 2 * rank)
    data0 = {k: v.to(dev) for k, v in batch.items()}
    gatherer = D.PoseGatherer(dev) if use_dist else None

    if rank == 0:
        tick("model + batch ready")
    dt, last, poses = measure(model, data0, args, use_dist, world, gatherer, prof)
    if rank == 0:
        tick("headline measured")
    graphed = len(model._graphs) > 0
    ev_steps = args.steps
    if graphed and prof is not None:
        # a replayed graph bypasses the Python-level launch wrappers: time the launches of ONE extra eager step
        mode_was, model.graph_mode = model.graph_mode, False
        prof.on = True
        model(dict(data0))
        torch.cuda.synchronize()
        prof.on = False
        model.graph_mode = mode_was
        ev_steps = 1
    if use_dist:
        assert poses is not None and poses[0].shape[0] == world * B, "gathered poses do not cover the global batch"
    ok = bool(torch.isfinite(last["R"]).all())
    if args.dump_poses and rank == 0:
        src = poses if poses is not None else (last["R"], last["t"], last["inliers"])
        torch.save({"R": src[0].cpu(), "t": src[1].cpu(), "inliers": src[2].cpu()}, args.dump_poses)

    out = None
    if rank == 0:
        roof = None
        if prof is not None and prof.records:
            stages, by = prof.summary(ev_steps)
            roof = roofline_entry(stages, by, B, args.dtype, model)
            if roof is not None and not args.lean:   # (--lean is what the rocprofv3 passes run: no probe kernel in their statistics)
                # what the socket power limit lets through the matrix pipe ALONE on this box right now (register-resident
                # pseudo-random bf16 operands, no memory traffic; mickey_hip_dev.h, LABNOTES R4.11): the ceiling of the MFMA
                # instruction itself next to the 2.5 PFLOP/s spec that `frac` is quoted against.  Never `peak`.
                try:
                    ps = ops.dev_mfma_sustained(dev)
                    roof["peak_sustained"], roof["frac_sustained"] = ps, roof["achieved"] / ps
                    roof["peak_sustained_note"] = ("back-to-back v_mfma_f32_16x16x32_bf16 on register-resident pseudo-random operands, "
                                                   "all CUs x 8 waves, one ~100-ms launch straight after the timed region "
                                                   "(mk_dev_mfma_sustained); zeros as operands: %.0f" % ops.dev_mfma_sustained(dev, zero_operands=True))
                except Exception as e:   # a probe must not cost the line
                    roof["peak_sustained_note"] = "probe failed: %s: %s" % (type(e).__name__, str(e)[:200])
            if roof is not None and graphed:
                roof["note"] = (roof.get("note", "") + "; forward replayed as a hipGraph in the timed region; kernel events "
                                "from one extra eager step").lstrip("; ")
        out = {
            "metric": "image pairs/sec (540x720)", "value": world * B * args.steps / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "MickeyRelativePose.forward, ViT-L/14 + 4 heads + dual-softmax + 20x100-hypothesis "
                                   "Procrustes RANSAC, %d pairs/GPU of 540x720 (W x H), random-init weights" % B,
                       "pairs_per_gpu": B, "global_batch": world * B, "image_hw": [H, W], "keypoints": 1938,
                       "hypotheses": 2000, "parallelism": "pairs sharded over %d GPU(s), 1 all-gather of poses "
                                                          "(side stream)" % world,
                       "hip_graph": graphed, "heads_operands": str(model.heads_dtype).replace("torch.", ""),
                       "affinity": affinity},
            "per_rank_ms_per_step": per_rank_ms(args.steps),
            "roofline": roof,
            "finite_output": ok,
        }

    single = rank == 0 and not use_dist
    if single and args.sustained:
        # the headline configuration over 60 steps: the part runs at its socket power limit and the clock it sustains
        # settles below what a 20-step region sees (round 1: -3 %)
        a3 = argparse.Namespace(**vars(args))
        a3.steps, a3.warmup = 60, 0
        dts, _, _ = measure(model, data0, a3, False, 1, None, None)
        out["sustained"] = {"value": B * a3.steps / dts, "unit": "pairs/s", "steps": a3.steps, "ms_per_step": dts / a3.steps * 1e3,
                            "what": "same model / batch / dtype, 60 back-to-back steps straight after the timed region"}
    if single and not args.no_single and B != 1:
        # BASELINE.json configs[1]: ONE 540x720 pair (latency; the forward is replayed as a hipGraph when AMD.GRAPH allows)
        d1 = {k: v.to(dev) for k, v in syn.synthetic_batch(B=1, H=H, W=W, seed=99).items()}
        a1 = argparse.Namespace(**vars(args))
        a1.steps, a1.warmup = 20, 5
        try:
            dt1, _, _ = measure(model, d1, a1, False, 1, None, None)
            out["single_pair"] = {"value": a1.steps / dt1, "unit": "pairs/s", "ms_per_pair": dt1 / a1.steps * 1e3, "steps": a1.steps,
                                  "hip_graph": len(model._graphs) > 0, "what": "BASELINE.json configs[1]: batch of one pair, same model"}
        except Exception as e:   # extra information: must not cost the headline line
            out["single_pair"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if single:
        tick("single pair measured")
    if single and args.include_h2d:
        from mickey_amd import input_pipeline as ip
        try:
            out["pcie_inclusive"] = ip.bench_h2d(model, B, H, W, steps=max(2, min(args.steps, 3)))
        except Exception as e:
            out["pcie_inclusive"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    del model
    torch.cuda.empty_cache()

    def leg(name, what, dtype, steps, warmup, batch_pairs=None, hw=(H, W), dominant="encoder_gemm", **mk):
        """One more full timed run (same measure(): barrier-free at N = 1, synchronize on both sides) of another
        precision / configuration, with its own per-stage roofline list (detail file)."""
        if not single or name not in args.leg_set:
            return
        try:
            _leg(name, what, dtype, steps, warmup, batch_pairs, hw, dominant, **mk)
        except Exception as e:   # a leg is extra information: its failure must not cost the headline line
            ops.attn_set_mode(args.attn_mode)
            torch.cuda.empty_cache()
            out.setdefault("legs", {})[name] = {"what": what, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    def _leg(name, what, dtype, steps, warmup, batch_pairs, hw, dominant, **mk):
        bp = batch_pairs or B
        attn_mode = mk.pop("attn_mode", None)   # dev knob (process-wide): set for this leg only
        natural = mk.pop("natural", False)      # 1/f-spectrum images generated on the device instead of white noise
        if attn_mode is not None:
            ops.attn_set_mode(attn_mode)
        m, _, _ = make_model(dtype, **mk)
        if natural:
            d = syn.natural_batch(B=bp, H=hw[0], W=hw[1], seed=1234, device=dev)
        else:
            d = {k: v.to(dev) for k, v in syn.synthetic_batch(B=bp, H=hw[0], W=hw[1], seed=1234).items()}
        a = argparse.Namespace(**vars(args))
        a.steps, a.warmup = steps, warmup
        if prof is not None:
            prof.records = []
        dtl, lastl, _ = measure(m, d, a, False, 1, None, prof)
        ent = {"what": what, "dtype": dtype,
               "heads_operands": "float32 as split fp16 planes" if m.heads_split else str(m.heads_dtype).replace("torch.", ""),
               "value": bp * steps / dtl, "unit": "pairs/s", "steps": steps, "warmup": warmup,
               "ms_per_step": dtl / steps * 1e3, "pairs_per_step": bp, "image_hw": list(hw),
               "finite_output": bool(torch.isfinite(lastl["R"]).all())}
        if prof is not None and prof.records and not len(m._graphs):
            st, _ = prof.summary(steps)
            ent["stages"] = [{k: s[k] for k in ("stage", "bound", "ms_per_step", "achieved", "peak", "unit", "frac", "hbm_frac") if k in s}
                             for s in st]
            dom = [s for s in st if s["stage"] == dominant]
            if dom:
                ent["roofline"] = {k: dom[0][k] for k in ("stage", "bound", "achieved", "peak", "unit", "frac")}
        del m, d
        torch.cuda.empty_cache()
        if attn_mode is not None:
            ops.attn_set_mode(args.attn_mode)
        out.setdefault("legs", {})[name] = ent

    leg("ref_split", "PRECISION-MATCHED: the reference's literal precision split (mickey_extractor.py:49-56): fp16 encoder + fp32 "
        "heads, the heads' 3x3 convolutions and linears on split fp16 operands (hi + lo planes: fp32-grade products; "
        "AMD.HEADS_DTYPE: split)", "fp16", max(5, args.steps // 2), 2, heads="split")
    leg("fp16", "fp16 operands EVERYWHERE (encoder as the reference's MICKEY.DINOV2.FLOAT16 mode, and the heads in fp16 too: "
        "less precise in the heads than the reference's split)", "fp16", args.steps, args.warmup)
    leg("ref_split_fp32mfma", "the reference's split with the heads on the exact fp32-input MFMA (AMD.HEADS_DTYPE: fp32; round 3's "
        "ref_split)", "fp16", 3, 1, heads="fp32")
    leg("attn_mfma16", "the headline configuration with the attention kernel on v_mfma_f32_16x16x32 (mk_attn_set_mode 5; LABNOTES "
        "R4.11 / R4.13)", args.dtype, args.steps, args.warmup, dominant="attention", attn_mode=5)
    leg("vit_small", "DINOv2 ViT-S/14 encoder (the size north_star names; 305 GFLOP per pair, attention 46 %% of it) "
        "+ the same heads / matcher / solver, %d pairs of 540x720" % B, args.dtype, max(5, args.steps // 2), 2,
        arch="vit_small", dominant="attention")
    leg("config5", "BASELINE.json configs[4]: 8 pairs of 1280x720 (51x91 grid, n = 4641), Sinkhorn matcher (10 "
        "iterations; governed by HBM: 20 LSE passes over the (n+1)^2 fp32 coupling matrix = 301 MB per pair at 540x720, "
        "1.72 GB here), fp16 operands", "fp16", 3, 1, batch_pairs=8, hw=(720, 1280), dominant="matcher", matcher="Sinkhorn")

    leg("natural", "REALISTIC OPERANDS (never `value`): the headline configuration on images with a 1/f amplitude spectrum "
        "(synthetic.natural_batch, drawn and filtered on the device) and encoder weights with the massive-activation statistics of "
        "released ViT-L weights (synthetic.plant_outliers: 4 residual channels at |x| ~ 600, compensating LayerNorm gains, hot "
        "attention heads) -- what the socket power limit lets through when the operand bits are not uniform noise (LABNOTES R4.11)",
        args.dtype, args.steps, args.warmup, outliers=True, natural=True)

    if rank == 0:
        tick("legs measured")
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu_baseline.threads_probe = bool(args.cpu_threads_probe)
                out["cpu_baseline"] = cpu_baseline(cfg, sd)
                tick("cpu baseline measured")
            except Exception as e:   # the line must still be printed (the contract's required keys stay present)
                out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": "failed",
                                       "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            if args.precision and getattr(cpu_baseline, "last_outputs", None):
                # precision of the timed configurations against the oracle outputs of THIS run (the pair the CPU leg just computed)
                try:
                    out["precision"] = precision_report(make_model, syn, dev, args, cpu_baseline.last_outputs)
                except Exception as e:   # a reporting leg must not lose the measured line
                    out["precision"] = {"error": "%s: %s" % (type(e).__name__, e)}
        else:
            out["cpu_baseline"] = None
        emit(out, args)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


LINE_LIMIT = 3072   # bytes; the driver parses the LAST stdout line out of an 8-KB tail (round 4's 21-KB line was cut: unparsed)


def compact_line(out, detail_path=None):
    """The ONE line rank 0 prints: the contract's keys + roofline + cpu_baseline + a <= 6-entry stage summary.  Pure function
    of the detail dict (unit-tested on CPU, tests/test_bench_cpu.py)."""
    def r4(x):
        return None if x is None else (round(x, 4) if isinstance(x, float) else x)

    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = r4(line["value"]), r4(line["ms_per_step"])
    if out.get("stub"):
        line["stub"] = True
    cfg = out.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "pairs_per_gpu", "global_batch", "image_hw", "hypotheses", "heads_operands",
                                          "parallelism", "hip_graph") if k in cfg}
    roof = out.get("roofline")
    if roof:
        line["roofline"] = {k: r4(roof.get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                         "algorithmic_bytes_per_launch", "avg_launch_ms", "launches", "traffic_source",
                                                         "peak_sustained", "frac_sustained") if k in roof or not k.endswith("sustained")}
        st = sorted(roof.get("stages") or [], key=lambda s: -s["ms_per_step"])[:6]
        line["stages"] = [{"stage": s["stage"], "ms_per_step": r4(s["ms_per_step"]), "frac": r4(s.get("frac"))} for s in st]
    else:
        line["roofline"] = None
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = ({k: r4(cb.get(k)) for k in ("value", "unit", "cores", "cores_available", "kind", "sample", "stage_seconds")
                             if k in cb} if cb else None)
    if (out.get("sustained") or {}).get("value") is not None:
        line["sustained_60"] = r4(out["sustained"]["value"])   # the headline configuration over 60 back-to-back steps (pairs/s)
    rs = (out.get("legs") or {}).get("ref_split") or {}
    if "value" in rs:
        line["value_ref_precision"] = r4(rs["value"])
        line["ref_precision"] = "fp16 ViT + fp32-grade heads (the reference's split, mickey_extractor.py:49-56)"
    nat = (out.get("legs") or {}).get("natural") or {}
    if "value" in nat:   # realistic operand statistics (1/f images, massive-activation weights): information, never `value`
        line["natural_operands"] = {"value": r4(nat["value"]), "encoder_gemm_tflops": r4((nat.get("roofline") or {}).get("achieved"))}
    sp = out.get("single_pair") or {}
    if "ms_per_pair" in sp:
        line["single_pair_ms"] = r4(sp["ms_per_pair"])
    pr = out.get("per_rank_ms_per_step")
    if pr and out.get("n_gpus", 1) > 1:
        line["per_rank_ms_per_step"] = {"min": r4(pr["min"]), "max": r4(pr["max"])}
    line["finite_output"] = out.get("finite_output")
    if detail_path:
        line["detail"] = detail_path
    txt = json.dumps(line, separators=(",", ":"))
    # cannot happen with the keys above; if it ever does, optional keys go first and free-text fields are cut -- the line is
    # printed whatever happens (an oversized or missing line is an unmeasured round)
    for k in ("ref_precision", "detail", "natural_operands", "stages", "single_pair_ms", "per_rank_ms_per_step", "sustained_60"):
        if len(txt) < LINE_LIMIT:
            break
        line.pop(k, None)
        txt = json.dumps(line, separators=(",", ":"))
    if len(txt) >= LINE_LIMIT:
        for obj, keys in ((line.get("roofline") or {}, ("traffic_source", "kernel")), (line.get("cpu_baseline") or {}, ("sample",)),
                          (line.get("config") or {}, ("workload", "parallelism"))):
            for k in keys:
                if isinstance(obj.get(k), str):
                    obj[k] = obj[k][:60]
        txt = json.dumps(line, separators=(",", ":"))
    return txt


def emit(out, args):
    """Detail -> file (+ a short summary on stderr); the compact line -> stdout, last."""
    path = None
    if args.detail != "none":
        path = args.detail or os.path.join("gpurun_out", "bench_detail.json")
        try:
            full = path if os.path.isabs(path) else os.path.join(ROOT, path)
            os.makedirs(os.path.dirname(full) or ".", exist_ok=True)
            with open(full, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as e:
            sys.stderr.write("bench.py: detail file not written (%s)\n" % e)
            path = None
    for name, ent in (out.get("legs") or {}).items():
        sys.stderr.write("bench.py leg %-20s %s\n" % (name, "%.1f pairs/s, %.2f ms/step" % (ent["value"], ent["ms_per_step"])
                                                      if "value" in ent else ent.get("error")))
    for st in ((out.get("roofline") or {}).get("stages") or []):
        sys.stderr.write("bench.py stage %-18s %8.3f ms/step  frac %s\n" % (st["stage"], st["ms_per_step"],
                                                                          "%.3f" % st["frac"] if st.get("frac") is not None else "-"))
    sys.stderr.flush()
    print(compact_line(out, path), flush=True)


if __name__ == "__main__":
    main()
