"""Configuration objects for the MicKey hot path.

The reference reads its config both by attribute (``cfg.PROCRUSTES.IT_RANSAC``,
reference lib/models/MicKey/modules/utils/probabilisticProcrustes.py:14-20) and by key
(``cfg['MICKEY']['DINOV2']``, reference lib/models/MicKey/modules/compute_correspondences.py:11-18)
on a yacs ``CfgNode``.  yacs is not a dependency of this package; ``CfgDict`` gives both access
styles on a plain dict, and ``as_cfg`` accepts a yacs CfgNode, a plain dict or a CfgDict.
"""
import copy
import os

import yaml


class CfgDict(dict):
    """dict with attribute access (read and write), nested."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return CfgDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def merge_from_file(self, path):
        with open(path, "r") as f:
            _merge(self, yaml.safe_load(f) or {})
        return self

    def merge_from_dict(self, other):
        _merge(self, other)
        return self


def _wrap(node):
    if isinstance(node, dict):
        return CfgDict({k: _wrap(v) for k, v in node.items()})
    return node


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = _wrap(v)


_DEFAULT_YAML = os.path.join(os.path.dirname(__file__), "configs", "mickey_default.yaml")


def default_cfg():
    """Hot-path defaults: same keys/values the reference ships in
    config/MicKey/curriculum_learning.yaml (MODEL, MICKEY, FEATURE_MATCHER, PROCRUSTES) and
    config/datasets/mapfree.yaml (DATASET.HEIGHT/WIDTH)."""
    with open(_DEFAULT_YAML, "r") as f:
        return _wrap(yaml.safe_load(f))


def load_cfg(*paths):
    cfg = default_cfg()
    for p in paths:
        cfg.merge_from_file(p)
    return cfg


def as_cfg(cfg):
    """Accept yacs CfgNode / dict / CfgDict; return a CfgDict with defaults filled in for keys
    the hot path reads (yacs nodes are dict subclasses, so the generic branch covers them)."""
    base = default_cfg()
    if cfg is None:
        return base
    if isinstance(cfg, dict):
        _merge(base, _strip_none(cfg))
        return base
    raise TypeError("cfg must be a dict-like config (yacs CfgNode, dict or CfgDict), got %r" % type(cfg))


def _strip_none(d):
    # the reference's yacs schema defaults every key to None (config/default.py:3-141); a None
    # must not overwrite a hot-path default
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out[k] = _strip_none(v)
        elif v is not None:
            out[k] = v
    return out
