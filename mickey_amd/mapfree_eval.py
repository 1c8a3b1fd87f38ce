"""Map-free evaluation harness: dataset on disk -> sharded GPU inference -> `submission.zip` -> the reference evaluator.

    python -m mickey_amd.mapfree_eval --dataset_path data/ --split val --checkpoint mickey.ckpt -o results/ \\
           [--config cfg.yaml] [--batch_size 32] [--evaluator_root /path/to/nianticlabs-mickey]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m mickey_amd.mapfree_eval ...   (8 GPUs)

What it replaces: the reference's `submission.py:70-99` (DataModule -> predict -> save_submission) for the val / test
splits, with the input pipeline of mickey_amd.input_pipeline (host cores decode only; resize, /255, CHW on the GPU) and
pairs sharded over the ranks (mickey_amd.distributed).  Scene parsing follows lib/datasets/mapfree.py:31-100,167-196
(intrinsics.txt, poses.txt, key-frame pairing with sample factor 5, intrinsics rescaled for the model resolution).

It takes PATHS and skips cleanly: when the dataset split, the checkpoint or (for the scoring step) the evaluator are not
there it says what is missing, prints `{"skipped": true, ...}` and exits 0 -- this image ships neither Map-free nor the
trained weights, so nothing here can be asserted about AUC; the harness is what turns those two files into the number
BASELINE.md quotes (VCRE AUC 0.74) on a machine that has them.
"""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

SAMPLE_FACTOR = {"train": 1, "val": 5, "test": 5}     # reference lib/datasets/mapfree.py:177


def read_intrinsics(scene_root, resize=None):
    """reference mapfree.py:31-49: `name fx fy cx cy W H` per line -> ({name: K at model resolution}, {name: K as stored}, (W, H))."""
    from .input_pipeline import correct_intrinsic_scale
    import torch
    Ks, K_ori, size = {}, {}, None
    with (Path(scene_root) / "intrinsics.txt").open("r") as f:
        for line in f.readlines():
            if "#" in line:
                continue
            parts = line.strip().split(" ")
            fx, fy, cx, cy, W, H = map(float, parts[1:])
            K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float32)
            K_ori[parts[0]] = K
            if resize is not None:
                K = correct_intrinsic_scale(torch.from_numpy(K), resize[0] / W, resize[1] / H).numpy()
            Ks[parts[0]] = K
            size = (int(W), int(H))
    return Ks, K_ori, size


def read_poses(scene_root):
    """reference mapfree.py:51-69: `name qw qx qy qz tx ty tz` -> {name: (q, t)} (world-to-camera)."""
    poses = {}
    with (Path(scene_root) / "poses.txt").open("r") as f:
        for line in f.readlines():
            if "#" in line:
                continue
            parts = line.strip().split(" ")
            qt = np.array(list(map(float, parts[1:])))
            poses[parts[0]] = (qt[:4], qt[4:])
    return poses


def scene_pairs(scene_root, poses, sample_factor, overlap_limits=(0.2, 0.7)):
    """reference mapfree.py:71-100: pre-computed overlaps when the scene has them (training scenes), otherwise the key
    frame seq0/frame_00000 against every `sample_factor`-th query frame of seq1."""
    ov = Path(scene_root) / "overlaps.npz"
    if ov.exists():
        f = np.load(ov, allow_pickle=True)
        idxs, overlaps = f["idxs"], f["overlaps"]
        if overlap_limits is not None:
            lo, hi = overlap_limits
            return idxs[(overlaps > lo) * (overlaps < hi)].copy()
        return idxs
    idxs = np.zeros((len(poses) - 1, 4), dtype=np.uint16)
    idxs[:, 2] = 1
    idxs[:, 3] = np.array([int(fn[-9:-4]) for fn in poses.keys() if "seq0" not in fn], dtype=np.uint16)
    return idxs[::sample_factor]


def scene_records(scene_root, resize, sample_factor):
    """One record per pair, in the reference's dataset order (mapfree.py:108-160), as mickey_amd.input_pipeline.PairFeeder
    takes them: image paths (decoded by the feeder's thread pool), the intrinsics OF THE STORED FRAMES (the feeder rescales
    them to the model resolution exactly as read_intrinsics(resize) does), scene_id, pair_names."""
    scene_root = Path(scene_root)
    poses = read_poses(scene_root)
    _, K_ori, _ = read_intrinsics(scene_root, None)
    recs = []
    for seqA, imgA, seqB, imgB in scene_pairs(scene_root, poses, sample_factor):
        a, b = f"seq{seqA}/frame_{imgA:05}.jpg", f"seq{seqB}/frame_{imgB:05}.jpg"
        recs.append({"image0": str(scene_root / a), "image1": str(scene_root / b), "K_color0": K_ori[a], "K_color1": K_ori[b],
                     "Kori_color0": K_ori[a], "Kori_color1": K_ori[b], "scene_id": scene_root.stem, "pair_names": (a, b)})
    return recs


def dataset_records(dataset_path, split, resize, scenes=None):
    root = Path(dataset_path) / split
    names = scenes if scenes else sorted(s.name for s in root.iterdir() if s.is_dir())
    recs = []
    for s in names:
        recs.extend(scene_records(root / s, resize, SAMPLE_FACTOR[split]))
    return recs


def missing_inputs(dataset_path, split, checkpoint):
    """What keeps the run from starting, as human-readable reasons (empty list: ready)."""
    why = []
    root = Path(dataset_path) / split if dataset_path else None
    if root is None or not root.is_dir():
        why.append("dataset split directory %s not found" % root)
    else:
        scenes = [s for s in root.iterdir() if s.is_dir()]
        if not scenes:
            why.append("no scene directories under %s" % root)
        elif not all((s / "intrinsics.txt").exists() and (s / "poses.txt").exists() for s in scenes):
            why.append("scene(s) without intrinsics.txt / poses.txt under %s" % root)
    if not checkpoint or not Path(checkpoint).is_file():
        why.append("checkpoint %s not found" % checkpoint)
    return why


def predict_to_zip(model, records, batch_size, resize, output_zip, sharded=False, device="cuda:0", rank=None, world=None):
    """records -> poses -> zip (reference submission.py:32-68).  With `sharded`, every rank runs this on the same record
    LIST but feeds (decodes, pins, uploads, preprocesses) only its contiguous slice of every global batch; one all-gather
    per batch returns all poses and rank 0 writes the file.  The Philox streams are keyed by the position in the global
    batch (`pair_base`), so the poses equal those of an unsharded run."""
    import torch
    from . import distributed as D
    from . import submission_io as sio
    from .input_pipeline import PairFeeder
    from collections import defaultdict
    results = defaultdict(list)
    if sharded:
        import torch.distributed as dist
        rank = dist.get_rank() if rank is None else rank
        world = dist.get_world_size() if world is None else world
    else:
        rank, world = 0, 1
    gbatches = [records[i:i + batch_size] for i in range(0, len(records), batch_size)]
    spans = [[D.shard_range(len(b), r, world) for r in range(world)] for b in gbatches]
    feeder = PairFeeder(None, batch_size, resize, device=device,
                        batches=[b[sp[rank][0]:sp[rank][1]] for b, sp in zip(gbatches, spans)])
    for data, gb, sp in zip(feeder, gbatches, spans):
        with torch.no_grad():
            if sharded:
                data["pair_base"] = sp[rank][0]
                R, t, inl = D.forward_local(model, data, [hi - lo for lo, hi in sp])
            else:
                R, t = model(data)
                inl = data["inliers"]
        sio.append_batch(results, [r["scene_id"] for r in gb], [r["pair_names"][1] for r in gb], R.detach().cpu().numpy(),
                         t.detach().cpu().numpy(), inl.detach().cpu().numpy())
    if rank == 0:   # the rank resolved above (argument, else the process group's; 0 when not sharded)
        sio.save_submission(results, output_zip)
    return results


def run_evaluator(evaluator_root, submission_zip, dataset_path, split):
    """The reference's own scorer (benchmark/mapfree.py:138-168) in a subprocess from its checkout.  Returns
    (metrics dict or None, message)."""
    root = Path(evaluator_root) if evaluator_root else None
    if root is None or not (root / "benchmark" / "mapfree.py").is_file():
        return None, "evaluator not found (pass --evaluator_root <checkout of nianticlabs/mickey>)"
    if split == "test":
        return None, "the test split has no public ground truth: upload the zip to the Map-free leaderboard"
    cmd = [sys.executable, "-m", "benchmark.mapfree", "--submission_path", str(Path(submission_zip).resolve()), "--split", split,
           "--dataset_path", str(Path(dataset_path).resolve())]
    env = dict(os.environ, PYTHONPATH=str(root))
    r = subprocess.run(cmd, cwd=str(root), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    start = r.stdout.find("{")
    if r.returncode != 0 or start < 0:
        return None, "evaluator did not produce metrics (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-400:])
    return json.loads(r.stdout[start:]), "ok"


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--dataset_path", default=None)
    ap.add_argument("--split", choices=("val", "test"), default="val")
    ap.add_argument("--checkpoint", default="")
    ap.add_argument("--config", default=None, help="yaml merged over the defaults (the reference's config/MicKey/*.yaml works)")
    ap.add_argument("--output_root", "-o", default="results/")
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--scenes", nargs="*", default=None)
    ap.add_argument("--evaluator_root", default=os.environ.get("MICKEY_REFERENCE_ROOT"))
    args = ap.parse_args(argv)

    why = missing_inputs(args.dataset_path, args.split, args.checkpoint)
    if why:
        for w in why:
            print("[mapfree_eval] skip: " + w, file=sys.stderr)
        print(json.dumps({"skipped": True, "reasons": why}))
        return 0
    import torch
    from .config import load_cfg
    from .model import build_model
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print(json.dumps({"skipped": True, "reasons": ["no GPU visible (mickey_amd has no CPU path)"]}))
        return 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
    cfg = load_cfg(*([args.config] if args.config else []))
    resize = (int(cfg["DATASET"]["WIDTH"]), int(cfg["DATASET"]["HEIGHT"]))
    model = build_model(cfg, args.checkpoint).to("cuda:%d" % local)
    records = dataset_records(args.dataset_path, args.split, resize, args.scenes)
    out_zip = Path(args.output_root) / "submission.zip"
    predict_to_zip(model, records, args.batch_size, resize, out_zip, sharded=world > 1, device="cuda:%d" % local)
    summary = {"skipped": False, "pairs": len(records), "submission": str(out_zip), "n_gpus": world}
    if int(os.environ.get("RANK", "0")) == 0:
        metrics, msg = run_evaluator(args.evaluator_root, out_zip, args.dataset_path, args.split)
        summary["evaluator"] = msg
        if metrics is not None:
            summary["metrics"] = metrics
        print(json.dumps(summary))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
