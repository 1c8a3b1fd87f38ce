"""Result sink of the evaluation flow: gathered poses -> the Map-free `submission.zip`.

Host code, mirrors the reference's `submission.py:17-68` (the `Pose` record, its text form, the per-scene files in the zip)
so that the evaluator (`benchmark/utils.py:18-78`, restated in `load_poses` for round-trip tests) reads byte-compatible
files.  `mat2quat` is transforms3d 0.4.1 (`resources/environment.yml:26` of the reference), which is not installed here:
its published algorithm (Bar-Itzhack 2000: eigenvector of the largest eigenvalue of a symmetric 4x4 matrix, w >= 0) is
restated below and cross-checked against scipy in the tests.
"""
from collections import defaultdict
from dataclasses import dataclass
from pathlib import Path
from zipfile import ZipFile

import numpy as np


def mat2quat(M):
    """Rotation matrix (any array with 9 elements, row-major; the reference passes [1,3,3] float32) -> (w, x, y, z).
    transforms3d 0.4.1 `quaternions.mat2quat`, expression for expression (so numpy's promotion rules decide the dtype
    exactly as they do there)."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M).flat
    K = np.array([
        [Qxx - Qyy - Qzz, 0, 0, 0],
        [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
        [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
        [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)               # uses the lower triangle only
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q *= -1
    return q


def _fields(values):
    """Space-separated fixed-point fields with 6 decimals: what numpy's array2string prints for the reference's
    formatter, without going through array2string."""
    return ' '.join('%.6f' % float(v) for v in np.asarray(values).reshape(-1))


@dataclass
class Pose:
    """One line of `pose_{scene}.txt`: `name qw qx qy qz tx ty tz conf` (reference submission.py:17-29).  The confidence is
    printed with Python's shortest round-trip float repr, as the reference's f-string does."""
    image_name: str
    q: np.ndarray
    t: np.ndarray
    inliers: float

    def __str__(self):
        return ' '.join((self.image_name, _fields(self.q), _fields(self.t), str(self.inliers)))


def append_batch(results_dict, scene_ids, query_names, R, t, inliers):
    """The per-batch body of the reference's `predict` (submission.py:42-60) on host arrays:
    R [B,3,3], t [B,1,3] or [B,3], inliers [B,1] or [B]; frames with a NaN/inf pose are skipped."""
    R = np.asarray(R, dtype=np.float32)
    t = np.asarray(t, dtype=np.float32).reshape(len(scene_ids), -1)
    inl = np.asarray(inliers).reshape(len(scene_ids), -1)
    for i, scene in enumerate(scene_ids):
        Ri, ti = R[i][None], t[i].reshape(-1)
        if np.isnan(Ri).any() or np.isnan(ti).any() or np.isinf(ti).any():
            continue
        results_dict[scene].append(Pose(image_name=query_names[i], q=mat2quat(Ri).reshape(-1), t=ti.reshape(-1),
                                        inliers=inl[i].reshape(-1)[0].item()))
    return results_dict


def predict(loader, model, to_device=None, sharded=False):
    """reference submission.py:32-61.  `sharded=True`: every rank calls this with the same loader; pairs are split over
    the ranks and one RCCL all-gather returns all poses to every rank (mickey_amd.distributed.forward_sharded)."""
    import torch
    results = defaultdict(list)
    for data in loader:
        if to_device is not None:
            data = to_device(data, model)
        with torch.no_grad():
            if sharded:
                from . import distributed as D
                R, t, inl = D.forward_sharded(model, data)
            else:
                R, t = model(data)
                inl = data['inliers']
        append_batch(results, data['scene_id'], data['pair_names'][1], R.detach().cpu().numpy(), t.detach().cpu().numpy(),
                     inl.detach().cpu().numpy())
    return results


def save_submission(results_dict, output_path):
    """submission.py:64-68: one `pose_{scene}.txt` per scene inside the zip, lines joined by '\\n' (no trailing newline)."""
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    with ZipFile(output_path, 'w') as zf:
        for scene, poses in results_dict.items():
            zf.writestr(f'pose_{scene}.txt', '\n'.join(str(p) for p in poses).encode('utf-8'))
    return output_path


# ---- the evaluator's reader (benchmark/utils.py:12-78), for round-trip tests -----------------------------------------
def _qinverse(q):
    w, x, y, z = q
    return np.array([w, -x, -y, -z]) / np.dot(q, q)


def _qmult(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2, w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2])


def _rotate_vector(v, q):
    return _qmult(q, _qmult(np.array([0.0, *v]), np.array([q[0], -q[1], -q[2], -q[3]])))[1:]


def load_poses(lines, load_confidence=True):
    """Text lines -> {frame number: (q_cam2world, camera centre, confidence)}; same validation and the same
    world2cam -> cam2world conversion as the evaluator."""
    expected = 9 if load_confidence else 8
    poses = {}
    for line in lines:
        parts = tuple(line.strip().split(' '))
        if len(parts) != expected or '#' in parts[0]:
            continue
        try:
            frame = int(parts[0][-9:-4])
            vals = tuple(map(float, parts[1:]))
        except ValueError:
            continue
        if any(np.isnan(v) or np.isinf(v) for v in vals):
            continue
        q = np.array(vals[0:4], dtype=np.float64)
        t = np.array(vals[4:7], dtype=np.float64)
        if np.isclose(np.linalg.norm(q), 0):
            continue
        qinv = _qinverse(q)
        poses[frame] = (qinv, -_rotate_vector(t, qinv), vals[7] if load_confidence else None)
    return poses
