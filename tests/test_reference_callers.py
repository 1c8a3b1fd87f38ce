"""The reference's OWN callers (demo_inference.py:79-138, submission.py:32-68) executed against the drop-in overlay
(BASELINE.json config #1: plumbing).  Needs /root/reference, so it runs in the build container only; there is no GPU
there, hence the forward itself is served by the CPU oracle inside the test process (see
tests/helpers/run_reference_callers.py) -- what is tested is the contract around the kernels."""
import json
import os
import subprocess
import sys
import zipfile

import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MICKEY_REFERENCE_ROOT", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lib", "models", "MicKey")), reason="reference tree not present")
def test_reference_demo_inference_and_submission_run_against_the_dropin(tmp_path):
    from PIL import Image
    from mickey_amd import synthetic as syn
    from mickey_amd.config import default_cfg
    # a MicKey-style checkpoint (no DINOv2 keys, as the reference saves them: model.py:291-298) + the hub file beside it,
    # tiny ViT so that the CPU oracle forward takes seconds; the reference's shipped YAML with the matching channel width
    cfg = default_cfg()
    cfg["MICKEY"]["DINOV2"]["CHANNEL_DIM"] = 128
    sd = syn.mickey_state_dict(cfg, seed=0, arch="vit_tiny_test")
    ckpt = {"state_dict": {k: v for k, v in sd.items() if "dinov2" not in k}, "epoch": 3, "global_step": 77}
    torch.save(ckpt, tmp_path / "mickey.ckpt")
    torch.save({k[len(syn.DINO_PREFIX):]: v for k, v in sd.items() if k.startswith(syn.DINO_PREFIX)}, tmp_path / "dinov2.pth")
    ref_yaml = yaml.safe_load(open(os.path.join(REF, "config", "MicKey", "curriculum_learning.yaml")))
    ref_yaml["MICKEY"]["DINOV2"]["CHANNEL_DIM"] = 128
    yaml.safe_dump(ref_yaml, open(tmp_path / "config.yaml", "w"))
    # the reference's toy pair, downscaled (the demo's own --resize path is exercised by leaving resize=None here and
    # giving it smaller files; intrinsics scaled accordingly)
    for i in (0, 1):
        Image.open(os.path.join(REF, "data", "toy_example", "im%d.jpg" % i)).resize((270, 360)).save(tmp_path / ("im%d.jpg" % i))
    with open(tmp_path / "intrinsics.txt", "w") as f:
        f.write("im0.jpg 274.85 274.85 134.33 175.92 270 360\nim1.jpg 274.53 274.53 134.43 175.92 270 360\n")
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + REF        # this repository BEFORE the reference
    env["MICKEY_DINOV2_WEIGHTS"] = str(tmp_path / "dinov2.pth")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "run_reference_callers.py"), "--workdir", str(tmp_path),
                        "--reference", REF], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    # the overlay resolved as INTEGRATION.md says: model factory from here, everything else from the reference
    assert out["builder_file"].startswith(ROOT) and out["visualization_file"].startswith(REF)
    # the demo wrote its four maps (score / depth per image) from what forward left in `data`
    names = sorted(os.path.basename(w[0]) for w in out["demo_written"])
    assert names == ["im0.depth.jpg", "im0.score.jpg", "im1.depth.jpg", "im1.score.jpg"], names
    for path, shape, _ in out["demo_written"]:
        assert shape[0] == 350 if "depth" in path else shape[0] == 360      # depth maps are grid * 14 = 25 * 14 rows
        assert os.path.getsize(path) > 1000
    # submission.predict consumed R, t, data['inliers'] for 2 batches x 2 pairs and its own writer produced the zip
    assert sorted(out["scenes"]) == ["s00000", "s00001"] and all(len(v) == 2 for v in out["scenes"].values())
    with zipfile.ZipFile(out["zip"]) as z:
        assert sorted(z.namelist()) == ["pose_s00000.txt", "pose_s00001.txt"]
        for n in z.namelist():
            for line in z.read(n).decode().splitlines():
                tok = line.split(" ")
                assert len(tok) == 9 and tok[0].startswith("seq1/frame_") and all(abs(float(x)) < 1e6 for x in tok[1:])
    # and the drop-in's own result sink reads the reference's file back (same wire format)
    from mickey_amd import submission_io as sio
    back = sio.read_submission(out["zip"]) if hasattr(sio, "read_submission") else None
    if back is not None:
        assert sorted(back) == ["s00000", "s00001"]
