"""Stand-ins for third-party modules the REFERENCE's callers import and this container lacks (cv2, yacs,
pytorch_lightning, transforms3d, torchvision, pyrender, trimesh).  TEST INFRASTRUCTURE: used only by
tests/helpers/run_reference_callers.py to execute the reference's own demo_inference.py / submission.py against the
drop-in `lib` overlay.  None of them touches hot-path arithmetic."""
import sys
import types

import numpy as np


def _cv2():
    from PIL import Image
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_COLOR, cv2.IMREAD_UNCHANGED = 1, -1
    cv2.COLOR_BGR2RGB, cv2.COLOR_RGB2BGR, cv2.COLOR_BGR2RGBA, cv2.COLOR_BGR2GRAY = 4, 4, 2, 6
    cv2.INTER_LINEAR = 1
    cv2.written = []

    def imread(path, flags=1):
        return np.ascontiguousarray(np.asarray(Image.open(str(path)).convert("RGB"))[:, :, ::-1])   # BGR like OpenCV

    def cvtColor(img, code):
        if code == 4:
            return np.ascontiguousarray(img[:, :, ::-1])
        if code == 2:
            a = np.full(img.shape[:2] + (1,), 255, img.dtype)
            return np.concatenate([img[:, :, 2::-1], a], -1)
        if code == 6:
            return (0.114 * img[:, :, 0] + 0.587 * img[:, :, 1] + 0.299 * img[:, :, 2]).astype(img.dtype)
        raise NotImplementedError(code)

    def resize(img, dsize, interpolation=1):
        w, h = int(dsize[0]), int(dsize[1])
        if img.dtype == np.uint8:
            return np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))
        chans = img if img.ndim == 3 else img[:, :, None]
        out = np.stack([np.asarray(Image.fromarray(chans[:, :, c].astype(np.float32), mode="F").resize((w, h), Image.BILINEAR))
                        for c in range(chans.shape[2])], -1)
        return out if img.ndim == 3 else out[:, :, 0]

    def addWeighted(a, alpha, b, beta, gamma):
        return np.clip(a.astype(np.float32) * alpha + b.astype(np.float32) * beta + gamma, 0, 255).astype(np.uint8)

    def imwrite(path, img):
        arr = np.asarray(img)
        cv2.written.append((str(path), arr.shape, str(arr.dtype)))
        arr = np.clip(arr, 0, 255).astype(np.uint8)
        if arr.ndim == 3 and arr.shape[2] >= 3:
            arr = arr[:, :, [2, 1, 0] + list(range(3, arr.shape[2]))]
        Image.fromarray(arr[:, :, :3] if arr.ndim == 3 else arr).save(str(path))
        return True

    cv2.imread, cv2.cvtColor, cv2.resize, cv2.addWeighted, cv2.imwrite = imread, cvtColor, resize, addWeighted, imwrite
    return cv2


def _yacs():
    import yaml

    class CfgNode(dict):
        def __init__(self, init=None):
            super().__init__()
            for k, v in (init or {}).items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

        def _merge(self, other):
            for k, v in other.items():
                if isinstance(v, dict) and isinstance(self.get(k), dict):
                    self[k]._merge(v)
                else:
                    self[k] = CfgNode(v) if isinstance(v, dict) else v

        def merge_from_file(self, path):
            with open(path) as f:
                self._merge(yaml.safe_load(f) or {})

        def clone(self):
            import copy
            return copy.deepcopy(self)

        def defrost(self):
            pass

        def freeze(self):
            pass

    yacs = types.ModuleType("yacs")
    cfgmod = types.ModuleType("yacs.config")
    cfgmod.CfgNode = CfgNode
    yacs.config = cfgmod
    return {"yacs": yacs, "yacs.config": cfgmod}


def install():
    import torch
    mods = {}
    if "cv2" not in sys.modules:
        mods["cv2"] = _cv2()
    try:
        import yacs  # noqa: F401
    except ImportError:
        mods.update(_yacs())
    try:
        import pytorch_lightning  # noqa: F401
    except ImportError:
        pl = types.ModuleType("pytorch_lightning")
        pl.LightningModule = torch.nn.Module
        pl.LightningDataModule = object
        mods["pytorch_lightning"] = pl
    try:
        import transforms3d  # noqa: F401
    except ImportError:
        from mickey_amd.submission_io import mat2quat   # restatement of transforms3d 0.4.1, pinned in test_submission_io_cpu.py
        t3 = types.ModuleType("transforms3d")
        q = types.ModuleType("transforms3d.quaternions")
        q.mat2quat = lambda M: mat2quat(np.asarray(M).reshape(3, 3))
        for nm in ("qinverse", "qmult", "rotate_vector", "quat2mat"):
            setattr(q, nm, lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("dataset path not exercised")))
        t3.quaternions = q
        mods.update({"transforms3d": t3, "transforms3d.quaternions": q})
    for name, attrs in (("torchvision", ()), ("torchvision.transforms", ("ColorJitter", "Grayscale")), ("pyrender", ()),
                        ("trimesh", ())):
        try:
            __import__(name)
        except ImportError:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, type(a, (), {"__init__": lambda self, *x, **y: None}))
            mods[name] = m
    if "torchvision" in mods and "torchvision.transforms" in mods:
        mods["torchvision"].transforms = mods["torchvision.transforms"]
    sys.modules.update(mods)
    import matplotlib.cm
    if not hasattr(matplotlib.cm, "get_cmap"):   # removed in matplotlib 3.9; the reference targets an older release
        import matplotlib
        matplotlib.cm.get_cmap = lambda name: matplotlib.colormaps[name]
    return mods
