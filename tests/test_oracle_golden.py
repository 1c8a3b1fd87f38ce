"""CPU: the oracle restatement (oracle/mickey_oracle.py) against fixtures produced by the reference
itself (tests/golden/*.npz, written by oracle/make_golden.py).  Weights/inputs are regenerated
from seeds.  This is what pins the oracle; the -m gpu tests then compare HIP against the oracle."""
import copy

import numpy as np
import torch

from mickey_amd import synthetic as syn
from oracle import mickey_oracle as O


def rel(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_vit_tiny_matches_reference(golden):
    g = golden("vit_tiny")
    D, depth, heads = syn.VIT_ARCH["vit_tiny_test"]
    sd = syn.dinov2_state_dict("vit_tiny_test", seed=3)
    img = torch.rand((2, 3, 84, 126), generator=torch.Generator().manual_seed(11))
    tok, pre = O.vit_forward_features(sd, "", img, heads, return_tokens=True)
    assert rel(tok, g["tokens"]) < 5e-6
    assert rel(pre, g["prenorm"]) < 5e-6
    assert rel(O.interp_pos_embed(sd["pos_embed"], 6, 9), g["pos"]) < 1e-6


def test_matcher_matches_reference(golden):
    g = golden("matcher")
    gen = torch.Generator().manual_seed(21)
    d0 = torch.nn.functional.normalize(torch.randn((2, 128, 150), generator=gen), dim=1)
    d1 = torch.nn.functional.normalize(torch.randn((2, 128, 131), generator=gen) + 0.5 * d0[:, :, :131], dim=1)
    ds = O.dual_softmax(d0, d1, 0.7, 0.1)
    assert rel(ds, g["dual_softmax"]) < 5e-6
    assert rel(O.sinkhorn(d0, d1, 1.3, 10), g["sinkhorn"]) < 5e-6
    # deterministic index output: bit-exact
    assert np.array_equal(O.mutual_nn_matches(torch.from_numpy(g["dual_softmax"])[:1]).numpy(), g["mnn"])


def test_solver_matches_reference(golden, cfg):
    g = golden("solver")
    scfg = copy.deepcopy(cfg)
    scfg["PROCRUSTES"]["IT_MATCHES"] = 4
    scfg["PROCRUSTES"]["IT_RANSAC"] = 25
    data, Rgt, tgt = syn.planted_pose_problem(B=3, h=14, w=12, seed=4321, angle_deg=(1.0, 1.5), t_norm=(0.03, 0.04))
    assert rel(Rgt, g["R_gt"]) == 0.0
    torch.manual_seed(5)
    R, t, conf, inl, dbg = O.estimate_pose(data, scfg, return_inliers=True, return_debug=True)
    # same RNG stream as the reference's two torch.multinomial calls => identical index sets
    assert np.array_equal(dbg["idx"].numpy(), g["idx"])
    assert np.array_equal(dbg["idx3"].numpy(), g["idx3"])
    assert np.array_equal(dbg["best"].numpy(), g["best"])
    assert rel(R, g["R"]) < 1e-5 and rel(t, g["t"]) < 5e-4 and rel(conf, g["conf"]) < 1e-4
    for b in range(3):
        assert inl[b].shape == g["inl%d" % b].shape
        assert rel(inl[b], g["inl%d" % b]) < 1e-4


def test_solver_zero_pose_on_invalid_matrix(cfg):
    scfg = copy.deepcopy(cfg)
    scfg["PROCRUSTES"]["IT_MATCHES"] = 2
    scfg["PROCRUSTES"]["IT_RANSAC"] = 5
    data, _, _ = syn.planted_pose_problem(B=2, h=14, w=12, seed=1)
    data["final_scores"] = torch.zeros_like(data["final_scores"])
    R, t, conf = O.estimate_pose(data, scfg)
    assert float(R.abs().sum()) == 0 and float(t.abs().sum()) == 0 and float(conf.abs().sum()) == 0


def test_full_forward_matches_reference(golden, cfg):
    """ViT-L extractor + matcher + solver at 182x196 (13x14 grid) against the reference's outputs."""
    g = golden("full_forward")
    sd = syn.mickey_state_dict(cfg, seed=0)
    data = syn.synthetic_batch(B=2, H=182, W=196, seed=1234)
    torch.manual_seed(77)
    with torch.no_grad():
        R, t = O.mickey_forward(sd, cfg, data, return_inliers=True)
    for k in ("kps0", "kps1", "depth_kp0", "depth_kp1", "scr0", "scr1", "dsc0", "dsc1", "scores", "kp_scores",
              "final_scores"):
        assert rel(data[k], g[k]) < 5e-5, k
    assert rel(R, g["R"]) < 1e-4 and rel(t, g["t"]) < 1e-4
    assert rel(data["inliers"], g["inliers"]) < 1e-4
    assert data["inliers_list"][0].shape == g["inliers_list0"].shape
    assert np.array_equal(O.mutual_nn_matches(torch.from_numpy(g["scores"])[:1]).numpy(), g["mnn"])


def test_noise_floor_fixture_is_consistent(golden):
    """noise_floor_lp.npz: the reference-mechanism fp16 floor at 182x196 is the one make_golden.py stores (two scripts, one
    reference run each), every entry is a positive finite scalar, and the orderings that must hold do: heads in bf16 add
    noise to an encoder in bf16; fp16 sits below bf16."""
    fl, old = golden("noise_floor_lp"), golden("noise_floor_fp16")
    for k in old:
        assert abs(float(fl["ref_fp16_182_" + k]) - float(old[k])) <= 1e-3 * float(old[k]), k
    assert all(np.isfinite(v) and float(v) > 0 for v in fl.values())
    for size in ("182", "720", "vits720"):
        for k in ("dsc0", "scr0", "scores", "final_scores"):
            assert fl["bf16_encheads_%s_%s" % (size, k)] > fl["bf16_enc_%s_%s" % (size, k)] > fl["fp16_enc_%s_%s" % (size, k)]


def test_committed_fixtures_reproduce_from_the_reference(tmp_path):
    """The oracle pin must stay runnable: regenerate every fixture from THE REFERENCE (oracle/make_golden.py and
    oracle/make_golden_train.py, in a subprocess so that the reference's `lib` namespace does not leak into this session) and compare with the committed
    files.  Skipped where /root/reference does not exist (the GPU box)."""
    import os
    import subprocess
    import sys
    import pytest
    from oracle import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # hot path; training-time RANSAC (row N3); the 16-bit noise floors the GPU tests assert against
    for script in ("make_golden.py", "make_golden_train.py", "make_noise_floor.py"):
        r = subprocess.run([sys.executable, os.path.join(root, "oracle", script), str(tmp_path)], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=1500, cwd=root)
        assert r.returncode == 0, r.stdout[-3000:]
    gold = os.path.join(root, "tests", "golden")
    names = sorted(f for f in os.listdir(gold) if f.endswith(".npz"))
    assert names and sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz")) == names
    for f in names:
        a, b = dict(np.load(os.path.join(gold, f))), dict(np.load(os.path.join(tmp_path, f)))
        assert sorted(a) == sorted(b), f
        for k in a:
            if f == "noise_floor_lp.npz":   # error norms of 16-bit CPU kernels: reproducible to the backend's blocking, not bitwise
                assert abs(float(b[k]) - float(a[k])) <= 0.05 * float(a[k]), (f, k, float(a[k]), float(b[k]))
            elif a[k].dtype.kind in "iuUSb":
                assert np.array_equal(a[k], b[k]), (f, k)          # index sets / text lines: bit-exact
            else:
                assert np.allclose(a[k], b[k], rtol=0, atol=0) or rel(b[k], a[k]) < 1e-6, (f, k, rel(b[k], a[k]))
