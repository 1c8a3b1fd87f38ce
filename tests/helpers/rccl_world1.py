"""Run under `python -m torch.distributed.run --nproc-per-node 1 ...` on a GPU box (tests/test_rccl_gpu.py): the RCCL path of
mickey_amd.distributed at world size 1 -- process-group init with backend nccl, PoseGatherer on a real side stream,
forward_sharded, and the sharded evaluation feed -- each compared with the un-distributed call.  Prints one JSON line."""
import json
import os
import sys
import tempfile
import zipfile

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from mickey_amd import distributed as D, mapfree_eval as ME, synthetic as syn
    from mickey_amd.config import default_cfg
    from mickey_amd.model import MickeyRelativePose
    from tests.helpers import tiny_mapfree
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    cfg = default_cfg()
    model = MickeyRelativePose(cfg)
    model.load_state_dict(syn.mickey_state_dict(cfg, seed=0))
    model = model.to(dev)
    batch = {k: v.to(dev) for k, v in syn.synthetic_batch(B=3, H=182, W=196, seed=5).items()}

    # 1. plain forward vs forward_sharded (all-gather over RCCL) vs the side-stream gatherer: bit-identical poses
    model.reseed(3)
    R0, t0 = model(dict(batch))
    model.reseed(3)
    R1, t1, c1 = D.forward_sharded(model, dict(batch))
    model.reseed(3)
    g = D.PoseGatherer(dev)
    d2 = dict(batch)
    R2l, t2l = model(d2)
    h = g.submit(R2l, t2l, d2["inliers"])
    junk = torch.randn((2048, 2048), device=dev) @ torch.randn((2048, 2048), device=dev)   # main stream keeps working
    R2, t2, c2 = g.wait(h)
    torch.cuda.synchronize()
    out["sharded_equal"] = bool(torch.equal(R0, R1) and torch.equal(t0, t1))
    out["gatherer_equal"] = bool(torch.equal(R0, R2) and torch.equal(t0, t2) and torch.equal(c1, c2))
    out["side_stream_is_not_current"] = g.stream is not None and g.stream != torch.cuda.current_stream(dev)
    out["junk_finite"] = bool(torch.isfinite(junk).all())

    # 2. ragged sizes path (padded all-gather) at world 1 + an explicit empty local slice
    Re, te, ce = D.forward_local(model, {"image0": batch["image0"][:0], "image1": batch["image1"][:0]}, [0])
    out["empty_slice_shapes"] = [list(Re.shape), list(te.shape), list(ce.shape)]

    # 3. the sharded evaluation feed == the unsharded one, byte for byte
    with tempfile.TemporaryDirectory() as tmp:
        tiny_mapfree.make(tmp, "val", scenes=("s00460", "s00461"), queries=11, size=(252, 336))
        recs = ME.dataset_records(tmp, "val", (252, 336))
        model.reseed(7)
        ME.predict_to_zip(model, recs, 4, (252, 336), os.path.join(tmp, "a.zip"), sharded=False, device=dev)
        model.reseed(7)
        ME.predict_to_zip(model, recs, 4, (252, 336), os.path.join(tmp, "b.zip"), sharded=True, device=dev)

        def body(p):
            with zipfile.ZipFile(p) as z:
                return {n: z.read(n) for n in sorted(z.namelist())}
        a, b = body(os.path.join(tmp, "a.zip")), body(os.path.join(tmp, "b.zip"))
        out["eval_zip_equal"] = a == b and len(a) == 2
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_WORLD1 " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
