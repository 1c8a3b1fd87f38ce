"""Generate tests/golden/*.npz from THE REFERENCE ITSELF and pin the oracle against it.

Run in the build container only (needs /root/reference):  python oracle/make_golden.py

For every stage it (1) runs the reference's own code on seeded weights/inputs, (2) runs the
oracle restatement (oracle/mickey_oracle.py) on the same inputs and asserts agreement, and
(3) stores the REFERENCE's outputs as fixtures.  Inputs/weights are not stored: they are regenerated
from seeds by mickey_amd.synthetic (bit-identical CPU generators), which keeps fixtures small.
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mickey_amd import synthetic as syn  # noqa: E402
from mickey_amd.config import default_cfg  # noqa: E402
from oracle import mickey_oracle as O  # noqa: E402
from oracle import ref_shim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(name, got, ref, tol):
    e = rel(got, ref)
    print("  %-28s rel-Fro %.3e (tol %.1e)" % (name, e, tol))
    assert e <= tol, name


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("  wrote %s.npz (%d arrays)" % (name, len(out)))


def gold_vit_tiny():
    print("[vit_tiny] reference DinoVisionTransformer(embed 128, depth 2, heads 2)")
    ref_shim.install()
    from lib.models.MicKey.modules.DINO_modules.dinov2 import DinoVisionTransformer
    from lib.models.MicKey.modules.DINO_modules.layers import MemEffAttention, NestedTensorBlock
    from functools import partial
    D, depth, heads = syn.VIT_ARCH["vit_tiny_test"]
    vit = DinoVisionTransformer(img_size=518, patch_size=14, embed_dim=D, depth=depth, num_heads=heads, mlp_ratio=4,
                                block_fn=partial(NestedTensorBlock, attn_class=MemEffAttention), init_values=1.0,
                                ffn_layer="mlp", block_chunks=0).eval()
    sd = syn.dinov2_state_dict("vit_tiny_test", seed=3)
    vit.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(11)
    img = torch.rand((2, 3, 84, 126), generator=g)
    with torch.no_grad():
        out = vit.forward_features(img)
        pos = vit.interpolate_pos_encoding(torch.zeros(1, 1 + 6 * 9, D), 84, 126)
    tok = O.vit_forward_features(sd, "", img, heads)
    check("x_norm_patchtokens", tok, out["x_norm_patchtokens"], 2e-6)
    check("interp_pos_embed", O.interp_pos_embed(sd["pos_embed"], 6, 9), pos, 1e-7)
    save("vit_tiny", tokens=out["x_norm_patchtokens"], prenorm=out["x_prenorm"], pos=pos)


def gold_full_forward(cfg):
    print("[full_forward] reference MickeyRelativePose (ViT-L, fp32) on 2 x 182x196 pairs")
    sd = syn.mickey_state_dict(cfg, seed=0)
    model = ref_shim.build_reference_model(cfg, sd, float16=False)
    batch = syn.synthetic_batch(B=2, H=182, W=196, seed=1234)
    data = {k: v.clone() for k, v in batch.items()}
    torch.manual_seed(77)
    with torch.no_grad():
        R, t = model(data, return_inliers=True)
    # oracle, same seed
    odata = {k: v.clone() for k, v in batch.items()}
    torch.manual_seed(77)
    with torch.no_grad():
        Ro, to = O.mickey_forward(sd, cfg, odata, return_inliers=True)
    for k in ("kps0", "kps1", "depth_kp0", "depth_kp1", "scr0", "scr1", "dsc0", "dsc1", "scores", "kp_scores",
              "final_scores"):
        check(k, odata[k], data[k], 5e-5)
    print("  R ref\n", R[0].numpy(), "\n  t ref", t[0].numpy(), "inliers", data["inliers"].flatten().numpy())
    check("R", Ro, R, 1e-4)
    check("t", to, t, 1e-4)
    check("inliers", odata["inliers"], data["inliers"], 1e-4)
    assert [tuple(x.shape) for x in odata["inliers_list"]] == [tuple(x.shape) for x in data["inliers_list"]]
    # mutual-NN (deterministic index output, B = 1)
    mnn = model.compute_matches.matcher.get_matches_list(data["scores"][:1])
    mo = O.mutual_nn_matches(data["scores"][:1])
    assert torch.equal(mnn.long(), mo.long()), "mutual-NN indices differ"
    print("  mutual-NN indices bit-exact (%d matches)" % len(mnn))
    save("full_forward", kps0=data["kps0"], kps1=data["kps1"], depth_kp0=data["depth_kp0"], depth_kp1=data["depth_kp1"],
         scr0=data["scr0"], scr1=data["scr1"], dsc0=data["dsc0"], dsc1=data["dsc1"], scores=data["scores"],
         kp_scores=data["kp_scores"], final_scores=data["final_scores"], R=R, t=t, inliers=data["inliers"],
         inliers_list0=data["inliers_list"][0], inliers_list1=data["inliers_list"][1], mnn=mnn.long())
    # fp16-encoder variant of the reference (its shipped default), for the low-precision noise floor
    model16 = ref_shim.build_reference_model(cfg, sd, float16=True)
    d16 = {k: v.clone() for k, v in batch.items()}
    with torch.no_grad():
        model16.compute_matches(d16)
    floor = {k: rel(d16[k], data[k]) for k in ("kps0", "depth_kp0", "scr0", "dsc0", "scores")}
    print("  fp16-encoder reference vs fp32 reference (noise floor):", {k: "%.2e" % v for k, v in floor.items()})
    save("noise_floor_fp16", **{k: np.float64(v) for k, v in floor.items()})
    return model


def gold_matcher(cfg, model):
    print("[matcher] reference dualSoftmax / sinkhorn / get_matches_list")
    from lib.models.MicKey.modules.utils.feature_matcher import sinkhorn as RefSinkhorn
    g = torch.Generator().manual_seed(21)
    d0 = torch.nn.functional.normalize(torch.randn((2, 128, 150), generator=g), dim=1)
    d1 = torch.nn.functional.normalize(torch.randn((2, 128, 131), generator=g) + 0.5 * d0[:, :, :131], dim=1)
    mm = model.compute_matches.matcher.matching_mat
    with torch.no_grad():
        mm.dustbin_score.fill_(0.7)
        ds = mm(d0, d1)
        sk_mod = RefSinkhorn(cfg["FEATURE_MATCHER"]["SINKHORN"])
        sk_mod.dustbin_score.fill_(1.3)
        sk = sk_mod(d0, d1, None)
        mnn = model.compute_matches.matcher.get_matches_list(ds[:1])
        mm.dustbin_score.fill_(1.0)
    check("dual_softmax", O.dual_softmax(d0, d1, 0.7, 0.1), ds, 2e-6)
    check("sinkhorn", O.sinkhorn(d0, d1, 1.3, 10), sk, 2e-6)
    assert torch.equal(O.mutual_nn_matches(ds[:1]).long(), mnn.long())
    save("matcher", dual_softmax=ds, sinkhorn=sk, mnn=mnn.long())


def gold_solver(cfg, model):
    print("[solver] reference estimate_pose_vectorized on a planted-pose problem")
    scfg = copy.deepcopy(cfg)
    scfg["PROCRUSTES"]["IT_MATCHES"] = 4
    scfg["PROCRUSTES"]["IT_RANSAC"] = 25
    ref_shim.install()
    from lib.models.MicKey.modules.utils.probabilisticProcrustes import e2eProbabilisticProcrustesSolver
    solver = e2eProbabilisticProcrustesSolver(scfg)
    data, Rgt, tgt = syn.planted_pose_problem(B=3, h=14, w=12, seed=4321, angle_deg=(1.0, 1.5), t_norm=(0.03, 0.04))
    torch.manual_seed(5)
    R, t, conf, inl = solver.estimate_pose_vectorized({k: v.clone() for k, v in data.items()}, return_inliers=True)
    torch.manual_seed(5)
    Ro, to, co, io, dbg = O.estimate_pose({k: v.clone() for k, v in data.items()}, scfg, return_inliers=True,
                                          return_debug=True)
    # same RNG stream => identical sampled index sets; floats differ only by BLAS summation order
    check("R", Ro, R, 1e-5)
    check("t", to, t, 2e-4)
    check("conf", co, conf, 1e-4)
    for a, b in zip(inl, io):
        assert a.shape == b.shape
        check("inliers_list", b, a, 1e-4)
    print("  oracle == reference; conf", conf.flatten().numpy(), "rounds", dbg["rounds"])
    print("  pose error vs planted: R %.2e  t %.2e" % (float((R - Rgt).norm(dim=(1, 2)).max()),
                                                      float((t - tgt).norm(dim=(1, 2)).max())))
    save("solver", R=R, t=t, conf=conf, idx=dbg["idx"].int(), idx3=dbg["idx3"].int(), best=dbg["best"].int(),
         R_best=dbg["R_best"], t_best=dbg["t_best"], score=dbg["score"], R_gt=Rgt, t_gt=tgt,
         inl0=inl[0], inl1=inl[1], inl2=inl[2])
    # degenerate input: an all-zero matching matrix -> torch.multinomial raises -> the reference's
    # except branch returns the all-zero pose for the whole batch (probabilisticProcrustes.py:331-336)
    bad = {k: v.clone() for k, v in data.items()}
    bad["final_scores"] = torch.zeros_like(bad["final_scores"])
    Rz, tz, cz = solver.estimate_pose_vectorized(bad)
    Rzo, tzo, czo = O.estimate_pose(bad, scfg)
    assert float(Rz.abs().sum()) == 0 and float(Rzo.abs().sum()) == 0 and float(czo.abs().sum()) == 0


def golden_submission_lines():
    """tests/golden/submission_lines.npz: text lines produced by the REFERENCE's own `Pose.__str__` (the dataclass is
    exec'd from submission.py:16-29; the module itself does not import here because transforms3d is absent)."""
    import re
    src = open(os.path.join(ref_shim.REF_ROOT if hasattr(ref_shim, "REF_ROOT") else "/root/reference", "submission.py")).read()
    m = re.search(r"@dataclass\nclass Pose:.*?\n\n\n", src, re.S)
    ns = {"np": np}
    exec("from dataclasses import dataclass\n" + m.group(0), ns)
    rng = np.random.default_rng(7)
    names, qs, ts, inl, lines = [], [], [], [], []
    for i in range(16):
        q = rng.normal(size=4).astype(np.float32)
        q /= np.linalg.norm(q)
        t = (rng.normal(size=3) * 10.0 ** rng.integers(-3, 2)).astype(np.float32)
        c = float(np.float32(rng.uniform(0, 400)))
        name = "seq1/frame_%05d.jpg" % (i * 37)
        names.append(name); qs.append(q); ts.append(t); inl.append(c)
        lines.append(str(ns["Pose"](image_name=name, q=q, t=t, inliers=c)))
    np.savez(os.path.join(GOLD, "submission_lines.npz"), names=np.array(names), q=np.stack(qs), t=np.stack(ts),
             inliers=np.array(inl, dtype=np.float64), lines=np.array(lines))
    print("submission_lines: %d lines, e.g. %s" % (len(lines), lines[0]))


def main(out_dir=None):
    """out_dir: write the fixtures somewhere else (tests/test_oracle_golden.py regenerates them into a temp dir and
    compares with the committed ones)."""
    global GOLD
    if out_dir is not None:
        GOLD = out_dir
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    cfg = default_cfg()
    gold_vit_tiny()
    model = gold_full_forward(cfg)
    gold_matcher(cfg, model)
    gold_solver(cfg, model)
    golden_submission_lines()
    ref_shim.uninstall()
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)

