"""Generate tests/golden/train_ransac.npz from THE REFERENCE's MetricPoseLoss and pin oracle/train_oracle.py against it.

Run in the build container only (needs /root/reference):  python oracle/make_golden_train.py [out_dir]

The reference class is imported as it is (lib/models/MicKey/modules/loss/loss_class.py); only torch.multinomial is
replaced while it runs, by a function that performs the real CPU draw from a seeded generator and records it, so the same
draws can be replayed into the oracle and into the HIP path.  Inputs are regenerated from seeds by
oracle.train_oracle.synthetic_batch, so the fixture holds only draws and outputs.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mickey_amd.config import _wrap  # noqa: E402
from oracle import ref_shim  # noqa: E402
from oracle import train_oracle as TO  # noqa: E402

# (name, B, n, seed, overrides of GENERATE_HYPOTHESES / SAMPLER / loss kind)
CASES = [
    ("small", 3, 40, 5, {"IT_MATCHES": 4, "IT_RANSAC": 5, "S": 64, "LOSS": "VCRE", "NOISE": 0.03}),
    ("pose_err", 2, 40, 6, {"IT_MATCHES": 3, "IT_RANSAC": 4, "S": 96, "LOSS": "POSE_ERR", "NOISE": 0.10}),
    ("default", 2, 64, 7, {"IT_MATCHES": 20, "IT_RANSAC": 20, "S": 512, "LOSS": "VCRE", "NOISE": 0.12}),
]


def case_cfg(ov):
    cfg = TO.default_loss_cfg()
    L = cfg["LOSS_CLASS"]
    L["GENERATE_HYPOTHESES"]["IT_MATCHES"] = ov["IT_MATCHES"]
    L["GENERATE_HYPOTHESES"]["IT_RANSAC"] = ov["IT_RANSAC"]
    L["SAMPLER"]["NUM_SAMPLES_MATCHES"] = ov["S"]
    L["LOSS_FUNCTION"] = ov["LOSS"]
    return _wrap(cfg)


def run_reference(batch, cfg, seed):
    ref_shim.install()
    if "transforms3d" not in sys.modules:   # imported by lib/benchmarks/reprojection.py for a function this path never calls
        t3 = types.ModuleType("transforms3d")
        q = types.ModuleType("transforms3d.quaternions")
        q.quat2mat = q.mat2quat = None
        t3.quaternions = q
        sys.modules["transforms3d"], sys.modules["transforms3d.quaternions"] = t3, q
    from lib.models.MicKey.modules.loss.loss_class import MetricPoseLoss
    loss = MetricPoseLoss(cfg)
    gen = torch.Generator().manual_seed(seed)
    draws = []
    real = torch.multinomial

    def recording(p, k, *a, **kw):
        out = real(p, k, generator=gen)
        draws.append(out.clone())
        return out

    torch.multinomial = recording
    try:
        b = {k: v.clone() for k, v in batch.items()}
        avg, outputs, grads, nvalid = loss.RANSAC_vectorized(b)
        avg.backward()
        b2 = {k: v.clone() for k, v in batch.items()}
        draws_keep = list(draws)
        it = iter(draws_keep)
        torch.multinomial = lambda p, k, *a, **kw: next(it)
        single = loss.single_iteration_RANSAC(b2, False)
    finally:
        torch.multinomial = real
        ref_shim.uninstall()
    return dict(avg_loss=avg.detach(), gradients=grads[0].detach(), nvalid=nvalid, mask_topk=outputs["mask_topk"],
                avg_loss_rot=outputs["avg_loss_rot"].detach(), avg_loss_trans=outputs["avg_loss_trans"].detach(),
                g_kps0=outputs["kps0"].grad, g_kps1=outputs["kps1"].grad, g_depth0=outputs["depth0"].grad,
                g_depth1=outputs["depth1"].grad, idx_outer=draws_keep[0], idx_inner=draws_keep[1],
                s_baseline=single[0].detach(), s_losses_rot=single[1].detach(), s_losses_trans=single[2].detach(),
                s_gradients=single[3].detach(), s_gradients_b=single[4].detach())


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main(out_dir=None):
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden")
    store = {}
    for name, B, n, seed, ov in CASES:
        cfg = case_cfg(ov)
        batch = TO.synthetic_batch(B, n, seed, noise=ov["NOISE"])
        ref = run_reference(batch, cfg, seed + 100)
        print("[%s] B=%d n=%d  it %dx%d  S=%d  %s: avg_loss %.6f  valid %d" % (
            name, B, n, ov["IT_MATCHES"], ov["IT_RANSAC"], ov["S"], ov["LOSS"], float(ref["avg_loss"]), ref["nvalid"]))
        # pin the oracle: same draws -> same everything
        avg, outputs, grads, nvalid, dbg = TO.ransac_vectorized(batch, cfg, ref["idx_outer"], ref["idx_inner"])
        avg.backward()
        pairs = [("avg_loss", avg.detach(), ref["avg_loss"], 1e-6), ("gradients", grads[0], ref["gradients"], 1e-6),
                 ("g_kps0", outputs["kps0"].grad, ref["g_kps0"], 1e-4), ("g_kps1", outputs["kps1"].grad, ref["g_kps1"], 1e-4),
                 ("g_depth0", outputs["depth0"].grad, ref["g_depth0"], 1e-4), ("g_depth1", outputs["depth1"].grad, ref["g_depth1"], 1e-4),
                 ("avg_loss_rot", outputs["avg_loss_rot"].detach(), ref["avg_loss_rot"], 1e-6),
                 ("avg_loss_trans", outputs["avg_loss_trans"].detach(), ref["avg_loss_trans"], 1e-6)]
        for nm, got, want, tol in pairs:
            e = rel(got, want)
            print("  %-16s rel-Fro %.3e (tol %.0e)" % (nm, e, tol))
            assert e <= tol, (name, nm, e)
        assert nvalid == ref["nvalid"]
        assert torch.equal(outputs["mask_topk"], ref["mask_topk"])
        s = TO.single_iteration(batch, cfg, ref["idx_outer"], ref["idx_inner"])
        assert torch.equal(s[4], ref["s_gradients_b"]), "gradients_b"
        assert rel(s[3], ref["s_gradients"]) <= 1e-6 and rel(s[0].detach(), ref["s_baseline"]) <= 1e-6
        print("  refinement rounds histogram:", torch.bincount(dbg["rounds"].long(), minlength=5).tolist())
        for k, v in ref.items():
            arr = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
            if k in ("gradients", "s_gradients", "s_gradients_b"):   # sparse: keep the non-zero cells only
                nz = np.flatnonzero(arr)
                store["%s/%s_nz" % (name, k)] = nz.astype(np.int64)
                store["%s/%s_val" % (name, k)] = arr.reshape(-1)[nz]
            else:
                store["%s/%s" % (name, k)] = arr
        store["%s/meta" % name] = np.array([B, n, seed, ov["IT_MATCHES"], ov["IT_RANSAC"], ov["S"], 0 if ov["LOSS"] == "VCRE" else 1,
                                           int(round(ov["NOISE"] * 1000))])
    np.savez_compressed(os.path.join(out_dir, "train_ransac.npz"), **store)
    print("wrote train_ransac.npz (%d arrays)" % len(store))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
