"""Map-free evaluation harness, host side: it takes paths and skips cleanly; its scene parsing yields the reference
dataset's pairs, names and intrinsics (compared with lib/datasets/mapfree.py run in a subprocess)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from mickey_amd import mapfree_eval as ME
from tests.helpers import tiny_mapfree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MICKEY_REFERENCE_ROOT", "/root/reference")


def test_skips_cleanly_without_dataset_or_checkpoint(tmp_path, capsys):
    assert ME.main(["--dataset_path", str(tmp_path / "nowhere"), "--checkpoint", str(tmp_path / "none.ckpt")]) == 0
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["skipped"] and len(out["reasons"]) == 2
    tiny_mapfree.make(str(tmp_path / "data"), "val", queries=3)
    assert ME.main(["--dataset_path", str(tmp_path / "data"), "--split", "val", "--checkpoint", ""]) == 0
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["skipped"] and out["reasons"] == ["checkpoint  not found"]
    os.makedirs(tmp_path / "data" / "val" / "broken_scene")
    assert any("intrinsics.txt" in w for w in ME.missing_inputs(str(tmp_path / "data"), "val", __file__))


def test_evaluator_leg_reports_instead_of_failing(tmp_path):
    metrics, msg = ME.run_evaluator(None, tmp_path / "s.zip", tmp_path, "val")
    assert metrics is None and "evaluator not found" in msg
    metrics, msg = ME.run_evaluator(str(tmp_path), tmp_path / "s.zip", tmp_path, "val")
    assert metrics is None and "evaluator not found" in msg
    (tmp_path / "benchmark").mkdir()
    (tmp_path / "benchmark" / "mapfree.py").write_text("raise SystemExit(3)\n")
    metrics, msg = ME.run_evaluator(str(tmp_path), tmp_path / "s.zip", tmp_path, "val")
    assert metrics is None and "rc 3" in msg
    (tmp_path / "benchmark" / "mapfree.py").write_text("import json; print(json.dumps({'AUC @ VCRE < 90px': 0.5}))\n")
    metrics, msg = ME.run_evaluator(str(tmp_path), tmp_path / "s.zip", tmp_path, "val")
    assert msg == "ok" and metrics == {"AUC @ VCRE < 90px": 0.5}
    assert ME.run_evaluator(str(tmp_path), tmp_path / "s.zip", tmp_path, "test")[0] is None


def test_records_follow_the_dataset_layout(tmp_path):
    split_root = tiny_mapfree.make(str(tmp_path), "val", scenes=("s1", "s0"), queries=11, size=(126, 168))
    recs = ME.dataset_records(str(tmp_path), "val", (126, 168))
    # scenes sorted; key frame against every 5th query frame: 0, 5, 10
    assert [r["scene_id"] for r in recs] == ["s0"] * 3 + ["s1"] * 3
    assert [r["pair_names"][1] for r in recs[:3]] == ["seq1/frame_00000.jpg", "seq1/frame_00005.jpg", "seq1/frame_00010.jpg"]
    assert all(r["pair_names"][0] == "seq0/frame_00000.jpg" and os.path.isfile(r["image0"]) and os.path.isfile(r["image1"]) for r in recs)
    assert recs[0]["K_color0"].dtype == np.float32 and recs[0]["K_color0"].shape == (3, 3)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lib", "datasets")), reason="reference tree not present")
def test_records_match_the_reference_dataset(tmp_path):
    import torch
    from mickey_amd.input_pipeline import correct_intrinsic_scale, decode_rgb
    split_root = tiny_mapfree.make(str(tmp_path), "test", queries=12, size=(126, 168))
    W, H = 126, 168          # same as stored: the reference's image tensors are then the decoded bytes / 255, no resampling
    env = dict(os.environ, PYTHONPATH=REF + os.pathsep + ROOT)   # reference first: `lib.datasets` is its own
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "run_reference_dataset.py"), split_root, str(W), str(H), "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    ref = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    recs = ME.dataset_records(str(tmp_path), "test", (W, H))
    assert len(recs) == len(ref) == 2 * 3
    for a, b in zip(recs, ref):
        assert a["scene_id"] == b["scene_id"] and list(a["pair_names"]) == b["pair_names"]
        assert np.array_equal(a["Kori_color0"], np.array(b["Kori_color0"], dtype=np.float32))
        img0 = decode_rgb(a["image0"]).astype(np.float64) / 255.0
        assert list(img0.transpose(2, 0, 1).shape) == b["image0_shape"]
        assert abs(float(np.float32(img0).astype(np.float64).sum()) - b["image0_sum"]) < 1e-2
    # intrinsics at a model resolution different from the stored one: the feeder's rescale == the reference's
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "run_reference_dataset.py"), split_root, "270", "360", "5"],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert r2.returncode == 0, r2.stdout[-3000:]
    ref2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for a, b in zip(recs, ref2):
        K = correct_intrinsic_scale(torch.from_numpy(a["K_color0"]), 270 / 126, 360 / 168).numpy()
        assert np.allclose(K, np.array(b["K_color0"], dtype=np.float32), rtol=0, atol=1e-4)
        Ks, _, size = ME.read_intrinsics(os.path.dirname(os.path.dirname(a["image0"])), (270, 360))
        assert np.allclose(Ks[a["pair_names"][0]], np.array(b["K_color0"], dtype=np.float32), rtol=0, atol=1e-4) and size == (126, 168)
