"""Row N1 end to end: JPEG files on disk -> PairFeeder (thread-pool decode into the pinned ring, H2D on a side stream,
resize / normalise kernel) -> forward, at the benchmark shape.  Reports pairs/s with the decode INCLUDED, for several
decode-pool sizes, next to the forward alone -- i.e. how many host cores one GPU needs.  (dev tool)

    python tools/bench_feeder.py [--pairs 32] [--batches 6] [--workers 8 16 32 64]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import synthetic as syn  # noqa: E402
from mickey_amd.config import default_cfg  # noqa: E402
from mickey_amd.input_pipeline import PairFeeder, decode_rgb  # noqa: E402
from mickey_amd.model import MickeyRelativePose  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--workers", type=int, nargs="*", default=[8, 16, 32, 64])
    args = ap.parse_args()
    from PIL import Image
    dev = torch.device("cuda:0")
    W, H = 540, 720
    tmp = tempfile.mkdtemp(prefix="mk_feeder_")
    g = np.random.default_rng(0)
    files = []
    for i in range(64):   # photo-like content (smooth + texture): JPEG decode time depends on it
        base = g.integers(0, 256, (H // 8, W // 8, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(base).resize((W, H), Image.BICUBIC)).astype(np.int16)
        img = np.clip(img + g.integers(-12, 13, img.shape), 0, 255).astype(np.uint8)
        path = os.path.join(tmp, "f%03d.jpg" % i)
        Image.fromarray(img).save(path, quality=90)
        files.append(path)
    kb = sum(os.path.getsize(f) for f in files) / len(files) / 1024
    t0 = time.perf_counter()
    for f in files[:16]:
        decode_rgb(f)
    dec_ms = (time.perf_counter() - t0) / 16 * 1e3
    K = np.array([[590.0, 0, 270.0], [0, 590.0, 360.0], [0, 0, 1.0]], dtype=np.float32)
    n = args.pairs * (args.batches + 1)
    recs = [{"image0": files[i % 64], "image1": files[(i * 7 + 3) % 64], "K_color0": K, "K_color1": K} for i in range(n)]
    cfg = default_cfg()
    cfg["AMD"]["ENCODER_DTYPE"] = "bf16"
    model = MickeyRelativePose(cfg)
    model.load_state_dict(syn.mickey_state_dict(cfg, seed=0))
    model = model.to(dev)
    out = {"what": "JPEG files -> decode pool -> pinned ring -> H2D -> resize kernel -> forward (row N1)", "pairs_per_batch": args.pairs,
           "jpeg_kib": round(kb, 1), "decode_ms_per_frame_one_core": round(dec_ms, 2), "host_cores": os.cpu_count(), "runs": []}
    # forward alone (inputs resident)
    data = next(iter(PairFeeder(recs[:args.pairs], args.pairs, (W, H), device=dev, workers=16)))
    for _ in range(2):
        model(dict(data))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.batches):
        model(dict(data))
    torch.cuda.synchronize()
    out["forward_alone_pairs_per_s"] = args.pairs * args.batches / (time.perf_counter() - t0)
    for w in args.workers:
        feeder = PairFeeder(recs, args.pairs, (W, H), device=dev, workers=w)
        it = iter(feeder)
        model(next(it))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = 0
        for d in it:
            model(d)
            m += d["image0"].shape[0]
        torch.cuda.synchronize()
        out["runs"].append({"decode_threads": w, "pairs_per_s": m / (time.perf_counter() - t0)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
