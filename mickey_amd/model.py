"""Drop-in replacement of the reference's inference module.

Same contract as reference lib/models/MicKey/compute_pose.py::MickeyRelativePose:
``forward(data, return_inliers=False) -> (R [B,3,3], t [B,1,3])`` with the reference's in-place side
effects on ``data``, ``on_load_checkpoint`` / ``load_state_dict`` with the reference's key names,
``.e2e_Procrustes.num_samples_matches``, ``.compute_matches.{extractor,matcher}``, ``.parameters()``.
All arithmetic of ``forward`` runs in libmickey_hip.so on the module's device; there is no CPU / ATen
fallback (a CPU-resident module raises on forward).
"""
import copy
import os

import torch
import torch.nn as nn

from . import _native, pipeline, weights
from .config import as_cfg
from .synthetic import DINO_PREFIX

_LP = {"bf16": torch.bfloat16, "fp16": torch.float16, "bfloat16": torch.bfloat16, "float16": torch.float16}


class _SolverView:
    """Attribute view of the PROCRUSTES config (reference probabilisticProcrustes.py:12-20)."""

    def __init__(self, cfg):
        P = cfg["PROCRUSTES"]
        self.it_RANSAC = P["IT_RANSAC"]
        self.it_matches = P["IT_MATCHES"]
        self.num_samples_matches = P["NUM_SAMPLED_MATCHES"]
        self.num_corr_3d_3d = P["NUM_CORR_3D_3D"]
        self.num_refinements = P["NUM_REFINEMENTS"]
        self.th_inlier = P["TH_INLIER"]
        self.th_soft_inlier = P["TH_SOFT_INLIER"]


class _MatcherView:
    def __init__(self, owner):
        self._owner = owner

    def get_matches_list(self, scores, min_conf=0.0):
        """Mutual nearest neighbours, sorted by score: reference feature_matcher.py:19-46 (B = 1 there;
        batched here, returns the B = 1 tensor for a batch of one, a list otherwise)."""
        from . import ops
        m, c = ops.mutual_nn(scores.contiguous())
        outs = [m[b, : int(c[b])].long() for b in range(scores.shape[0])]
        return outs[0] if len(outs) == 1 else outs


class _ComputeMatchesView:
    def __init__(self, owner):
        self.matcher = _MatcherView(owner)
        self.extractor = owner
        self.down_factor = owner.cfg["MICKEY"]["DINOV2"]["DOWN_FACTOR"]
        self.dsc_dim = owner.cfg["MICKEY"]["DSC_HEAD"]["LAST_DIM"]


class MickeyRelativePose(nn.Module):
    def __init__(self, cfg, dinov2_weights=None):
        super().__init__()
        self.cfg = as_cfg(cfg)
        amd = self.cfg["AMD"]
        self.lp_dtype = _LP[str(amd.get("ENCODER_DTYPE", "bf16")).lower()]
        self.lean = bool(amd.get("LEAN", False))
        self.seed = int(amd.get("SEED", 0))
        # hipGraph replay of the whole forward for launch-bound batches: "auto" (<= GRAPH_MAX_IMAGES images), True, False
        self.graph_mode = amd.get("GRAPH", "auto")
        self.graph_max_images = int(amd.get("GRAPH_MAX_IMAGES", 8))
        self._graphs = {}
        self._calls = 0
        self._ctr = None   # device-resident 2 * _calls: the Philox stream offset (read by the kernels, so a graph can advance it)
        # a single registered parameter carries the module's device (reference callers use
        # next(model.parameters()).device, lib/utils/data.py:7); the real weights live in _sd
        self._anchor = nn.Parameter(torch.zeros(1), requires_grad=False)
        self._sd = {}
        self._dev_weights = None
        self._ws = pipeline.Workspace()
        self.e2e_Procrustes = _SolverView(self.cfg)
        object.__setattr__(self, "compute_matches", _ComputeMatchesView(self))
        if dinov2_weights is None:
            path = amd.get("DINOV2_WEIGHTS") or os.environ.get("MICKEY_DINOV2_WEIGHTS")
            if path:
                dinov2_weights = torch.load(path, map_location="cpu")
        if dinov2_weights is not None:
            for k, v in dinov2_weights.items():
                self._sd[DINO_PREFIX + k] = v.detach().cpu()
        self.eval()

    # ---- checkpoint contract ---------------------------------------------------------------------
    def state_dict(self, *args, **kwargs):
        return dict(self._sd)

    def load_state_dict(self, state_dict, strict=True):
        self._sd = {k: v.detach().cpu() for k, v in state_dict.items() if torch.is_tensor(v)}
        self._dev_weights = None
        if strict and not any(k.startswith(DINO_PREFIX) for k in self._sd):
            raise RuntimeError("state_dict holds no DINOv2 weights (%s*): pass dinov2_weights= / AMD.DINOV2_WEIGHTS or "
                               "call on_load_checkpoint first, as the reference's build_model does" % DINO_PREFIX)
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def on_load_checkpoint(self, checkpoint):
        """MicKey checkpoints are saved without the frozen DINOv2 weights; re-inject them before
        load_state_dict (reference compute_pose.py:39-48)."""
        for k, v in self._sd.items():
            if "dinov2" in k:
                checkpoint["state_dict"][k] = v

    def is_eval_model(self, is_eval):  # reference compute_pose.py:50-60; inference-only here
        if not is_eval:
            raise NotImplementedError("mickey_amd implements the inference path only")

    def _apply(self, fn, *a, **k):
        self._dev_weights = None
        self._graphs = {}
        self._ctr = None
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return self._anchor.device

    def device_weights(self):
        dev = self._anchor.device
        if dev.type != "cuda":
            raise _native.MickeyHipError("MickeyRelativePose.forward needs the module on a GPU (model.cuda()); "
                                         "mickey_amd has no CPU fallback")
        if self._dev_weights is None:
            if not self._sd:
                raise RuntimeError("no weights loaded")
            self._dev_weights = weights.prepare(self._sd, self.cfg, dev, self.lp_dtype)
        return self._dev_weights

    # ---- forward ---------------------------------------------------------------------------------
    @torch.no_grad()
    def compute_correspondences(self, data):
        """reference compute_correspondences.py:52-92 + compute_pose.py:23: fills the data dict."""
        W = self.device_weights()
        dev = self._anchor.device
        im0 = data["image0"].to(device=dev, dtype=torch.float32)
        im1 = data["image1"].to(device=dev, dtype=torch.float32)
        B = im0.shape[0]
        same = im0.shape == im1.shape
        imgs = [torch.cat([im0, im1], 0)] if same else [im0, im1]   # one 2B-image pass when shapes agree
        outs = []
        for im in imgs:
            feat, gh, gw = pipeline.encoder_forward(W, self._ws, im.contiguous())
            scr, kps, depth, dsc = pipeline.heads_forward(W, self._ws, feat, im.shape[0], gh, gw, self.cfg)
            outs.append((scr, kps, depth, dsc, gh, gw))
        if same:
            scr, kps, depth, dsc, gh, gw = outs[0]
            parts = [(scr[:B], kps[:B], depth[:B], dsc[:B], gh, gw), (scr[B:], kps[B:], depth[B:], dsc[B:], gh, gw)]
        else:
            parts = outs
        for i, (scr, kps, depth, dsc, gh, gw) in enumerate(parts):
            data["kps%d_shape" % i] = [gh, gw]
            data["depth%d_map" % i] = depth.reshape(B, 1, gh, gw)
            data["kps%d" % i] = kps.contiguous()
            data["depth_kp%d" % i] = depth.contiguous()
            data["scr%d" % i] = scr.contiguous()
            data["dsc%d" % i] = dsc.contiguous()
        data["down_factor"] = self.cfg["MICKEY"]["DINOV2"]["DOWN_FACTOR"]
        scores, kp, fin = pipeline.match(W, self.cfg, data["dsc0"], data["dsc1"], data["scr0"], data["scr1"], self.lean)
        if scores is not None:
            data["scores"] = scores
            data["kp_scores"] = kp
        data["final_scores"] = fin
        return data["kps0"], data["dsc0"], data["kps1"], data["dsc1"]

    @torch.no_grad()
    def estimate_pose(self, data, return_inliers=False):
        dev = self._anchor.device
        K0 = data["K_color0"].to(device=dev, dtype=torch.float32).contiguous()
        K1 = data["K_color1"].to(device=dev, dtype=torch.float32).contiguous()
        # offset of this call's Philox streams = 2 * (number of forwards so far), kept on the device
        if self._ctr is None:
            self._ctr = torch.full((1,), 2 * self._calls, device=dev, dtype=torch.int64)
        from . import ops
        self._calls += 1
        ops.counter_add(self._ctr, 2)
        sol = pipeline.solve(self.cfg, data["final_scores"], data["kps0"], data["depth_kp0"], data["kps1"], data["depth_kp1"],
                             K0, K1, seed=self.seed, offset=0, offset_dev=self._ctr)
        if return_inliers:
            return sol["R"], sol["t"], sol["inliers"], pipeline.inliers_list(sol)
        return sol["R"], sol["t"], sol["inliers"]

    def _forward_eager(self, data, return_inliers=False):
        try:
            self.compute_correspondences(data)
            res = self.estimate_pose(data, return_inliers)
        except _native.MickeyHipError:
            raise  # a missing / failing HIP library is never papered over
        if return_inliers:
            data["inliers_list"] = res[3]
        data["R"], data["t"], data["inliers"] = res[0], res[1], res[2]
        return res[0], res[1]

    def reseed(self, seed=None, calls=0):
        """Restart the on-device sampler streams: call number `calls` + 1 comes next (optionally with a new Philox seed).
        Two modules with equal weights, seed and call count produce bit-identical poses."""
        if seed is not None:
            if int(seed) != self.seed:
                self._graphs = {}   # the seed is a kernel argument baked into captured graphs
            self.seed = int(seed)
        self._calls = int(calls)
        if self._ctr is not None:
            self._ctr.fill_(2 * self._calls)

    _GRAPH_INPUTS = ("image0", "image1", "K_color0", "K_color1")

    def _wants_graph(self, data, return_inliers):
        mode = self.graph_mode
        if mode is False or str(mode).lower() in ("false", "0", "off") or return_inliers:
            return False
        if self._anchor.device.type != "cuda" or not all(torch.is_tensor(data.get(k)) for k in self._GRAPH_INPUTS):
            return False
        if data["image0"].shape != data["image1"].shape:
            return False
        if mode is True or str(mode).lower() in ("true", "1", "on"):
            return True
        return 2 * data["image0"].shape[0] <= self.graph_max_images   # "auto": the launch-bound regime

    def _forward_graphed(self, data):
        """One hipGraph per input signature: ~330 kernel launches replayed with a single submission (a single 540x720
        pair: 7.9 -> 7.3 ms).  Inputs are copied into static buffers and everything written into `data` is cloned out of
        the graph's memory pool, so results never alias a later call."""
        dev = self._anchor.device
        key = tuple((k, tuple(data[k].shape)) for k in self._GRAPH_INPUTS)
        entry = self._graphs.get(key)
        if entry is None:
            static = {k: torch.empty(data[k].shape, device=dev, dtype=torch.float32) for k in self._GRAPH_INPUTS}
            for k in self._GRAPH_INPUTS:
                static[k].copy_(data[k])
            # warm-up outside the capture (lazy weight preparation, workspace allocation, kernel attributes) without
            # consuming random-stream positions
            calls = self._calls
            self._forward_eager(dict(static))
            self._calls = calls
            self._ctr.fill_(2 * calls)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            gdata = dict(static)
            with torch.cuda.graph(graph):
                self._forward_eager(gdata)
            self._calls = calls
            entry = (graph, static, gdata)
            self._graphs[key] = entry
        graph, static, gdata = entry
        for k in self._GRAPH_INPUTS:
            static[k].copy_(data[k], non_blocking=True)
        self._calls += 1
        graph.replay()
        for k, v in gdata.items():
            if k not in self._GRAPH_INPUTS:
                data[k] = v.clone() if torch.is_tensor(v) else copy.copy(v)
        return data["R"], data["t"]

    def forward(self, data, return_inliers=False):
        if self._wants_graph(data, return_inliers):
            return self._forward_graphed(data)
        return self._forward_eager(data, return_inliers)


def build_model(cfg, checkpoint="", dinov2_weights=None):
    """reference lib/models/builder.py:5-18."""
    cfg = as_cfg(cfg)
    if cfg["MODEL"] != "MicKey":
        raise NotImplementedError()
    model = MickeyRelativePose(cfg, dinov2_weights=dinov2_weights)
    ckpt = torch.load(checkpoint, map_location="cpu", weights_only=False)  # Lightning checkpoints are pickles
    ckpt = {"state_dict": dict(ckpt["state_dict"])}
    model.on_load_checkpoint(ckpt)
    model.load_state_dict(ckpt["state_dict"])
    if torch.cuda.is_available():
        model = model.cuda()
    model.eval()
    return model
