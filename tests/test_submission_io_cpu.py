"""the pose wire format and submission.zip writer against the reference's own text lines
(tests/golden/submission_lines.npz, produced by submission.py's Pose.__str__) and the evaluator's reader."""
import io
import zipfile
from collections import defaultdict

import numpy as np
import pytest


def test_pose_lines_match_reference(golden):
    from mickey_amd.submission_io import Pose
    g = golden("submission_lines")
    for name, q, t, c, line in zip(g["names"], g["q"], g["t"], g["inliers"], g["lines"]):
        assert str(Pose(image_name=str(name), q=q, t=t, inliers=float(c))) == str(line)


def test_mat2quat_against_scipy_and_conventions():
    from scipy.spatial.transform import Rotation
    from mickey_amd.submission_io import mat2quat
    rs = Rotation.random(200, random_state=3)
    for R in rs.as_matrix():
        q = mat2quat(R[None].astype(np.float32))            # the reference passes [1,3,3] float32
        assert q[0] >= 0
        x, y, z, w = Rotation.from_matrix(R).as_quat()
        ref = np.array([w, x, y, z])
        ref = ref if ref[0] >= 0 else -ref
        assert np.abs(q - ref).max() < 2e-6
    assert np.allclose(mat2quat(np.eye(3)), [1, 0, 0, 0])
    # 180-degree turn about z: w = 0, sign fixed by the eigenvector routine, still a unit quaternion of that rotation
    q = mat2quat(np.diag([-1.0, -1.0, 1.0]))
    assert abs(abs(q[3]) - 1) < 1e-12 and abs(q[0]) < 1e-12


def test_submission_zip_round_trip(tmp_path):
    from scipy.spatial.transform import Rotation
    from mickey_amd.submission_io import append_batch, load_poses, save_submission
    rng = np.random.default_rng(0)
    results = defaultdict(list)
    truth = {}
    for b in range(3):                                       # three batches of four pairs, two scenes, one NaN pose
        R = Rotation.random(4, random_state=b).as_matrix().astype(np.float32)
        t = rng.normal(size=(4, 1, 3)).astype(np.float32)
        inl = rng.uniform(0, 300, size=(4, 1)).astype(np.float32)
        if b == 1:
            t[2] = np.nan
        scenes = ["s%05d" % (i % 2) for i in range(4)]
        names = ["seq1/frame_%05d.jpg" % (b * 10 + i) for i in range(4)]
        append_batch(results, scenes, names, R, t, inl)
        for i in range(4):
            if not np.isnan(t[i]).any():
                truth[(scenes[i], b * 10 + i)] = (R[i], t[i].reshape(3), float(inl[i, 0]))
    path = save_submission(results, tmp_path / "out" / "submission.zip")
    seen = 0
    with zipfile.ZipFile(path) as zf:
        assert sorted(zf.namelist()) == ["pose_s00000.txt", "pose_s00001.txt"]
        for fn in zf.namelist():
            text = zf.read(fn).decode("utf-8")
            assert not text.endswith("\n")
            poses = load_poses(io.StringIO(text).readlines(), load_confidence=True)
            for frame, (q_c2w, centre, conf) in poses.items():
                R, t, c = truth[(fn[5:-4], frame)]
                seen += 1
                assert abs(conf - c) < 1e-4
                assert np.abs(centre - (-R.T.astype(np.float64) @ t)).max() < 1e-4       # camera centre = -R^T t
                Rc2w = Rotation.from_quat([q_c2w[1], q_c2w[2], q_c2w[3], q_c2w[0]]).as_matrix()
                assert np.abs(Rc2w - R.T).max() < 1e-4
    assert seen == len(truth) == 11


def test_predict_with_stub_model(tmp_path):
    """The reference's predict loop (submission.py:32-61) on a stub model: batches are walked in order, `inliers` is read
    from the data dict the model filled, NaN poses are dropped, scenes are grouped."""
    import torch
    from scipy.spatial.transform import Rotation
    from mickey_amd.submission_io import load_poses, predict, save_submission

    class Stub(torch.nn.Module):
        def forward(self, data):
            B = len(data["scene_id"])
            R = torch.from_numpy(Rotation.random(B, random_state=int(data["seed"])).as_matrix()).float()
            t = torch.full((B, 1, 3), float(data["seed"]))
            if data["seed"] == 2:
                t[0] = float("nan")
            data["inliers"] = torch.arange(B, dtype=torch.float32).reshape(B, 1) + 10 * data["seed"]
            return R, t

    loader = [{"scene_id": ["s00001", "s00002", "s00001"], "seed": s,
               "pair_names": (["seq0/frame_00000.jpg"] * 3, ["seq1/frame_%05d.jpg" % (10 * s + i) for i in range(3)])}
              for s in (1, 2, 3)]
    moved = []
    res = predict(loader, Stub(), to_device=lambda d, m: (moved.append(1), d)[1])
    assert len(moved) == 3 and sorted(res) == ["s00001", "s00002"]
    assert [p.image_name[-9:-4] for p in res["s00001"]] == ["00010", "00012", "00022", "00030", "00032"]   # 00020 was NaN
    assert [p.inliers for p in res["s00002"]] == [11.0, 21.0, 31.0]
    path = save_submission(res, tmp_path / "submission.zip")
    import zipfile
    with zipfile.ZipFile(path) as zf:
        lines = zf.read("pose_s00002.txt").decode().split("\n")
    assert len(lines) == 3 and set(load_poses(lines)) == {11, 21, 31}
