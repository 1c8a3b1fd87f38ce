"""Seeded synthetic weights and inputs for the MicKey hot path.

There are no pretrained weights or Map-free data on the build/GPU boxes, so parity tests and
``bench.py`` run on seeded random weights laid out EXACTLY like a reference checkpoint
(key names / shapes: SURVEY.md §8(b) "Checkpoint contract", probed from the reference's own
``state_dict()``), so the same dict loads into the reference model (``load_state_dict(strict=True)``),
into the CPU oracle and into the HIP path.

Generation uses an explicit CPU ``torch.Generator`` in a fixed key order, so the weights are
bit-identical on every box with the same torch build.  BatchNorm running stats and the dustbin
score are randomised (identity BN would make head tests weak).
"""
import math

import torch

VIT_ARCH = {
    # name: (embed_dim, depth, heads)   (reference DINO_modules/dinov2.py:306-345)
    "vit_tiny_test": (128, 2, 2),
    "vit_small": (384, 12, 6),
    "vit_base": (768, 12, 12),
    "vit_large": (1024, 24, 16),
}

HEADS = ("det_head", "det_offset", "depth_head", "dsc_head")
EXTRACTOR_PREFIX = "compute_matches.extractor."
DINO_PREFIX = EXTRACTOR_PREFIX + "dinov2_vitl14."
DUSTBIN_KEY = "compute_matches.matcher.matching_mat.dustbin_score"


_SHAPES_ONLY = False   # expected_keys(): enumerate names / shapes without drawing 350 M random numbers


def _randn(g, shape, std=1.0, mean=0.0):
    if _SHAPES_ONLY:
        return torch.empty(shape, device="meta")
    return torch.randn(shape, generator=g, dtype=torch.float32) * std + mean


def _rand(g, shape, lo, hi):
    if _SHAPES_ONLY:
        return torch.empty(shape, device="meta")
    return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo


def dinov2_state_dict(arch="vit_large", seed=0, prefix="", pos_grid=37):
    """Random DINOv2 ViT/14 weights with the hub file's key layout (reference
    DINO_modules/dinov2.py:88-150).  ``pos_grid``=37 is img_size 518 / 14."""
    D, depth, _ = VIT_ARCH[arch]
    g = torch.Generator().manual_seed(1000 + seed)
    sd = {}
    sd[prefix + "cls_token"] = _randn(g, (1, 1, D), 0.02)
    sd[prefix + "pos_embed"] = _randn(g, (1, 1 + pos_grid * pos_grid, D), 0.02)
    sd[prefix + "mask_token"] = torch.zeros(1, D)
    sd[prefix + "patch_embed.proj.weight"] = _randn(g, (D, 3, 14, 14), 1.0 / math.sqrt(588.0))
    sd[prefix + "patch_embed.proj.bias"] = _randn(g, (D,), 0.02)
    for i in range(depth):
        p = prefix + "blocks.%d." % i
        sd[p + "norm1.weight"] = _randn(g, (D,), 0.1, 1.0)
        sd[p + "norm1.bias"] = _randn(g, (D,), 0.02)
        sd[p + "attn.qkv.weight"] = _randn(g, (3 * D, D), 1.0 / math.sqrt(D))
        sd[p + "attn.qkv.bias"] = _randn(g, (3 * D,), 0.02)
        sd[p + "attn.proj.weight"] = _randn(g, (D, D), 1.0 / math.sqrt(D))
        sd[p + "attn.proj.bias"] = _randn(g, (D,), 0.02)
        sd[p + "ls1.gamma"] = _rand(g, (D,), 0.05, 0.3)
        sd[p + "norm2.weight"] = _randn(g, (D,), 0.1, 1.0)
        sd[p + "norm2.bias"] = _randn(g, (D,), 0.02)
        sd[p + "mlp.fc1.weight"] = _randn(g, (4 * D, D), 1.0 / math.sqrt(D))
        sd[p + "mlp.fc1.bias"] = _randn(g, (4 * D,), 0.02)
        sd[p + "mlp.fc2.weight"] = _randn(g, (D, 4 * D), 1.0 / math.sqrt(4 * D))
        sd[p + "mlp.fc2.bias"] = _randn(g, (D,), 0.02)
        sd[p + "ls2.gamma"] = _rand(g, (D,), 0.05, 0.3)
    sd[prefix + "norm.weight"] = _randn(g, (D,), 0.1, 1.0)
    sd[prefix + "norm.bias"] = _randn(g, (D,), 0.02)
    return sd


OUTLIER_LEVELS = (250.0, -400.0, 520.0, -600.0)


def plant_outliers(sd, arch="vit_large", prefix=DINO_PREFIX, seed=0, ln_gain=36.0, qk_gain=3.0):
    """Give random DINOv2 weights the activation statistics released ViT-L weights are known for (massive activations: a
    handful of residual channels carrying |x| in the hundreds in every token from an early block on, LayerNorm gains that
    compensate, a few attention heads with logits of several tens) -- the numerics a 16-bit residual stream, a LayerNorm folded
    into GEMM epilogues and a softmax without a per-tile maximum have to survive (tests/test_outliers_gpu.py).  In place, and
    deterministic in (arch, seed):
      * block `start` = depth // 4 writes OUTLIER_LEVELS into 4 channels of every token (mlp.fc2.bias, ls2.gamma = 1);
        later blocks add a token-dependent part on top (their fc2 rows x 40, ls2.gamma 0.3): |x| ~ 200 ... 700 there;
      * from then on the row variance is the outliers' (std ~ 29, the ordinary channels' 0.8): every later LayerNorm (norm1 / norm2 / final norm) gets
        weight x ln_gain on the ordinary channels and 0.02 on the outlier channels, as trained networks do;
      * every third block from `start` on, heads 1 and heads - 2 get their q and k rows (weight and bias) x qk_gain:
        attention logits of +-40 and more in those heads.
    Returns (channels, start)."""
    D, depth, heads = VIT_ARCH[arch]
    g = torch.Generator().manual_seed(3000 + seed)
    ch = torch.randperm(D, generator=g)[:len(OUTLIER_LEVELS)]
    start = max(1, depth // 4)
    level = torch.tensor(OUTLIER_LEVELS)
    ordinary = torch.ones(D, dtype=torch.bool)
    ordinary[ch] = False

    def ln(key):
        w = sd[key + ".weight"]
        w[ordinary] *= ln_gain
        w[ch] = 0.02
    for i in range(start, depth):
        p = prefix + "blocks.%d." % i
        if i == start:
            sd[p + "mlp.fc2.bias"][ch] = level
            sd[p + "ls2.gamma"][ch] = 1.0
        else:
            ln(p + "norm1")
            sd[p + "mlp.fc2.weight"][ch] *= 40.0
            sd[p + "ls2.gamma"][ch] = 0.3
            ln(p + "norm2")
        if i > start and (i - start) % 3 == 0:
            for h in (1, heads - 2):
                for base in (0, D):   # q rows, k rows of attn.qkv
                    rows = slice(base + h * 64, base + (h + 1) * 64)
                    sd[p + "attn.qkv.weight"][rows] *= qk_gain
                    sd[p + "attn.qkv.bias"][rows] *= qk_gain
    ln(prefix + "norm")
    return ch, start


def _basic_block(g, sd, p, cin, cout):
    sd[p + "conv1.weight"] = _randn(g, (cout, cin, 3, 3), math.sqrt(2.0 / (9 * cin)))
    sd[p + "conv2.weight"] = _randn(g, (cout, cout, 3, 3), math.sqrt(2.0 / (9 * cout)))
    for bn in ("bn1.", "bn2."):
        sd[p + bn + "weight"] = _rand(g, (cout,), 0.5, 1.5)
        sd[p + bn + "bias"] = _randn(g, (cout,), 0.1)
        sd[p + bn + "running_mean"] = _randn(g, (cout,), 0.1)
        sd[p + bn + "running_var"] = _rand(g, (cout,), 0.5, 1.5)
        sd[p + bn + "num_batches_tracked"] = torch.tensor(100, dtype=torch.long)
    if cin != cout:
        sd[p + "shortcut.0.weight"] = _randn(g, (cout, cin, 1, 1), math.sqrt(1.0 / cin))


def heads_state_dict(cfg, seed=0, prefix=EXTRACTOR_PREFIX):
    """Random head weights with the reference's key layout (reference mickey_extractor.py:67-251,
    extractor_utils.py:12-26, att_layers/transformer_utils.py:14-38)."""
    g = torch.Generator().manual_seed(2000 + seed)
    mk = cfg["MICKEY"]
    cin = mk["DINOV2"]["CHANNEL_DIM"]
    dims = list(mk["KP_HEADS"]["BLOCKS_DIM"])
    sd = {}
    for head in HEADS:
        hp = prefix + head + "."
        last = mk["DSC_HEAD"]["LAST_DIM"] if head == "dsc_head" else dims[3]
        chain = [cin, dims[0], dims[1], dims[2], last]
        for b in range(4):
            _basic_block(g, sd, hp + "resblock%d." % (b + 1), chain[b], chain[b + 1])
        d = dims[2]
        for l in range(3):
            lp = hp + "att_layer.layers.%d." % l
            for nm in ("q_proj", "k_proj", "v_proj", "merge"):
                sd[lp + nm + ".weight"] = _randn(g, (d, d), 1.0 / math.sqrt(d))
            sd[lp + "mlp.0.weight"] = _randn(g, (2 * d, 2 * d), 1.0 / math.sqrt(2 * d))
            sd[lp + "mlp.2.weight"] = _randn(g, (d, 2 * d), 1.0 / math.sqrt(2 * d))
            for nm in ("norm1", "norm2"):
                sd[lp + nm + ".weight"] = _randn(g, (d,), 0.1, 1.0)
                sd[lp + nm + ".bias"] = _randn(g, (d,), 0.05)
        if head == "det_head":
            sd[hp + "score.weight"] = _randn(g, (1, last, 1, 1), 4.0)
            sd[hp + "eps"] = torch.tensor(1e-16)
            sd[hp + "offset_par1"] = torch.tensor(0.5)
            sd[hp + "offset_par2"] = torch.tensor(2.0)
            sd[hp + "ones_kernel"] = torch.ones(1, 1, 3, 3)
        elif head == "det_offset":
            sd[hp + "xy_offset.weight"] = _randn(g, (2, last, 1, 1), 0.3)
        elif head == "depth_head":
            sd[hp + "depth.weight"] = _randn(g, (1, last, 1, 1), 0.3).abs() if not _SHAPES_ONLY else _randn(g, (1, last, 1, 1))
    return sd


def mickey_state_dict(cfg, seed=0, arch="vit_large", dustbin=1.0, outliers=False):
    """Full synthetic checkpoint ``state_dict`` (DINOv2 keys included).  outliers: plant_outliers() on the encoder."""
    sd = {}
    sd.update(dinov2_state_dict(arch, seed, prefix=DINO_PREFIX))
    if outliers and not _SHAPES_ONLY:
        plant_outliers(sd, arch, DINO_PREFIX, seed)
    sd.update(heads_state_dict(cfg, seed))
    if cfg["FEATURE_MATCHER"]["TYPE"] == "DualSoftmax":
        if cfg["FEATURE_MATCHER"]["DUAL_SOFTMAX"]["USE_DUSTBIN"]:
            sd[DUSTBIN_KEY] = torch.tensor(float(dustbin))
    else:
        sd[DUSTBIN_KEY] = torch.tensor(float(dustbin))
    return sd


def expected_keys(cfg, arch="vit_large"):
    """{key: shape} of a full MicKey state_dict for this config (the checkpoint contract of SURVEY.md 8(b)); used by
    MickeyRelativePose.load_state_dict to report missing / unexpected keys the way nn.Module's strict load does."""
    global _SHAPES_ONLY
    _SHAPES_ONLY = True
    try:
        sd = mickey_state_dict(cfg, 0, arch)
    finally:
        _SHAPES_ONLY = False
    return {k: tuple(v.shape) for k, v in sd.items()}


# ---------------------------------------------------------------------------------------------
# inputs

TOY_K0 = [[549.7018, 0.0, 268.6665], [0.0, 549.7018, 351.8357], [0.0, 0.0, 1.0]]
TOY_K1 = [[549.0616, 0.0, 268.8559], [0.0, 549.0616, 351.8485], [0.0, 0.0, 1.0]]


def synthetic_batch(B=1, H=720, W=540, seed=1234):
    """SURVEY.md §8(d) throughput batch: uniform images in [0,1), toy-example intrinsics
    (reference data/toy_example/intrinsics.txt)."""
    g0 = torch.Generator().manual_seed(seed)
    g1 = torch.Generator().manual_seed(seed + 1)
    return {
        "image0": torch.rand((B, 3, H, W), generator=g0),
        "image1": torch.rand((B, 3, H, W), generator=g1),
        "K_color0": torch.tensor(TOY_K0).repeat(B, 1, 1),
        "K_color1": torch.tensor(TOY_K1).repeat(B, 1, 1),
    }


def natural_batch(B=1, H=720, W=540, seed=1234, device=None):
    """synthetic_batch() with images that have a photograph's statistics instead of white noise: Gaussian noise shaped to a 1/f
    amplitude spectrum per channel (a shared luminance field + 30 % of an independent field per colour channel), each image
    rescaled to mean 0.45 / std 0.22 and clamped to [0, 1].  Neighbouring pixels are correlated (the patch embedding sees
    smooth patches, not 588 independent uniforms); used by bench.py's `natural` leg, never by the headline.  device: where
    the fields are drawn and filtered (input generation, not part of any forward)."""
    device = torch.device(device or "cpu")

    def field(g, n):
        x = torch.randn((n, H, W), generator=g, device=device)
        fy = torch.fft.fftfreq(H, device=device).view(H, 1)
        fx = torch.fft.rfftfreq(W, device=device).view(1, W // 2 + 1)
        amp = 1.0 / torch.sqrt(fy * fy + fx * fx).clamp_min(1.0 / max(H, W))
        amp[0, 0] = 0.0
        y = torch.fft.irfft2(torch.fft.rfft2(x) * amp, s=(H, W))
        return y / y.flatten(1).std(1).view(n, 1, 1)

    out = synthetic_batch(B, 8, 8, seed)   # the intrinsics
    for key, sd in (("image0", seed), ("image1", seed + 1)):
        g = torch.Generator(device=device).manual_seed(sd + 77)
        lum = field(g, B).view(B, 1, H, W)
        col = field(g, 3 * B).view(B, 3, H, W)
        img = lum + 0.3 * col
        img = img / img.flatten(1).std(1).view(B, 1, 1, 1)
        out[key] = (0.45 + 0.22 * img).clamp_(0.0, 1.0).contiguous()
    return {k: v.to(device) for k, v in out.items()}


def planted_pose_problem(B=1, h=51, w=38, seed=4321, inlier_frac=0.6, down=14, angle_deg=(5.0, 10.0),
                         t_norm=(0.3, 0.4)):
    """Solver-realism generator: a known relative pose is planted into
    keypoints/depths and a peaked ``final_scores`` matrix, so a solver that is statistically
    equivalent to the reference must recover it.  Returns (data dict, R_gt [B,3,3], t_gt [B,1,3])."""
    g = torch.Generator().manual_seed(seed)
    n = h * w
    K0 = torch.tensor(TOY_K0)
    K1 = torch.tensor(TOY_K1)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    grid = torch.stack([xs.reshape(-1), ys.reshape(-1)], 0)  # [2,n] (x, y)
    data = {k: [] for k in ("kps0", "kps1", "depth_kp0", "depth_kp1", "final_scores")}
    Rs, ts = [], []
    for b in range(B):
        ang = math.radians(angle_deg[0] + angle_deg[1] * (b % 3))
        axis = _randn(g, (3,))
        axis = axis / axis.norm()
        Kx = torch.tensor([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = torch.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * (Kx @ Kx)
        t = _randn(g, (3,))
        t = t / t.norm() * (t_norm[0] + t_norm[1] * (b % 3))
        kps0 = (grid + _rand(g, (2, n), 0, 1)) * down
        d0 = _rand(g, (n,), 1.0, 8.0)
        uv1 = torch.cat([kps0, torch.ones(1, n)], 0)
        X0 = d0 * (torch.linalg.inv(K0) @ uv1)  # [3,n]
        X1 = R @ X0 + t[:, None]
        proj = K1 @ X1
        uv = proj[:2] / proj[2:].clamp_min(1e-6)
        cell = torch.floor(uv / down)
        valid = (X1[2] > 0.1) & (cell[0] >= 0) & (cell[0] < w) & (cell[1] >= 0) & (cell[1] < h)
        keep = valid & (torch.rand(n, generator=g) < inlier_frac)
        j = (cell[1] * w + cell[0]).long().clamp(0, n - 1)
        # image-1 keypoints / depths: background random, planted cells consistent with (R, t)
        kps1 = (grid + _rand(g, (2, n), 0, 1)) * down
        d1 = _rand(g, (n,), 1.0, 8.0)
        i_idx = torch.nonzero(keep).flatten()
        # one source per destination cell (last writer wins, then rebuild the list consistently)
        owner = torch.full((n,), -1, dtype=torch.long)
        owner[j[i_idx]] = i_idx
        jj = torch.nonzero(owner >= 0).flatten()
        ii = owner[jj]
        kps1[:, jj] = uv[:, ii]
        d1[jj] = X1[2, ii]
        fs = _rand(g, (n, n), 0.0, 1e-9)
        fs[ii, jj] = 1e-6 * (0.5 + torch.rand(len(ii), generator=g))
        data["kps0"].append(kps0)
        data["kps1"].append(kps1)
        data["depth_kp0"].append(d0[None])
        data["depth_kp1"].append(d1[None])
        data["final_scores"].append(fs)
        Rs.append(R)
        ts.append(t[None])
    out = {k: torch.stack(v, 0).contiguous() for k, v in data.items()}
    out["K_color0"] = K0.repeat(B, 1, 1)
    out["K_color1"] = K1.repeat(B, 1, 1)
    return out, torch.stack(Rs, 0), torch.stack(ts, 0)
