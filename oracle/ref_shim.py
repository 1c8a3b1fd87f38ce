"""Import the reference (nianticlabs/mickey) in-process from /root/reference.

TEST INFRASTRUCTURE ONLY, and only usable in the build container (the GPU box has no
/root/reference).  Four shims, none touching hot-path arithmetic:
cv2 and pytorch_lightning stub modules, a dict-with-attributes cfg (yacs is absent) and a
``torch.hub.load_state_dict_from_url`` replacement that returns seeded DINOv2 weights.
"""
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("MICKEY_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "models", "MicKey"))


def _purge_lib():
    for name in [m for m in sys.modules if m == "lib" or m.startswith("lib.")]:
        del sys.modules[name]


def install():
    """Make `import lib....` resolve to the REFERENCE.  Both the reference's `lib/` and this repository's drop-in `lib/`
    are namespace packages (no __init__.py), so whichever root comes first on sys.path provides a module that both have
    (lib.models.builder, lib.models.MicKey.compute_pose): the reference root goes to the front and cached `lib*` modules
    are dropped.  uninstall() restores the drop-in."""
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    while REF_ROOT in sys.path:
        sys.path.remove(REF_ROOT)
    sys.path.insert(0, REF_ROOT)
    _purge_lib()
    for name, mod in _stub_modules().items():
        sys.modules.setdefault(name, mod)


def uninstall():
    while REF_ROOT in sys.path:
        sys.path.remove(REF_ROOT)
    _purge_lib()


def _stub_modules():
    out = {}
    if "cv2" not in sys.modules:
        out["cv2"] = types.ModuleType("cv2")
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")
        pl.LightningModule = torch.nn.Module
        out["pytorch_lightning"] = pl
    return out


def build_reference_model(cfg, state_dict, float16=False):
    """Reference MickeyRelativePose in eval mode with ``state_dict`` loaded strictly."""
    install()
    import copy
    cfg = copy.deepcopy(cfg)
    cfg["MICKEY"]["DINOV2"]["FLOAT16"] = bool(float16)
    dino = {k.split("dinov2_vitl14.", 1)[1]: v for k, v in state_dict.items() if "dinov2_vitl14." in k}
    old = torch.hub.load_state_dict_from_url
    torch.hub.load_state_dict_from_url = lambda *a, **k: dino
    try:
        from lib.models.MicKey.compute_pose import MickeyRelativePose
        model = MickeyRelativePose(cfg)
    finally:
        torch.hub.load_state_dict_from_url = old
    sd = {k: (v.half() if (float16 and "dinov2" in k and v.is_floating_point()) else v) for k, v in state_dict.items()}
    model.load_state_dict(sd, strict=True)
    return model.eval()
