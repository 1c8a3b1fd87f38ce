"""Checkpoint -> device weight images for the HIP kernels.

Input is a flat ``state_dict`` with the reference's key names; output is a
``DeviceWeights`` bag of contiguous device tensors laid out the way the kernels consume them:
  * ViT linears as [out, in] 16-bit matrices (nn.Linear's own layout: K contiguous for both MFMA
    operands), biases / LayerScale / LayerNorm vectors in fp32;
  * the patch-embedding conv as a [D, 640] matrix (588 real columns, zero padded to the K tile);
  * the heads' 3x3 convs as [Cout, 9*Cin (+ Cshortcut)] matrices in tap-major order with eval-mode
    BatchNorm folded in (reference utils/extractor_utils.py:18-35: y = (conv(x) - mean) * w /
    sqrt(var + 1e-5) + b), the four heads stacked along a leading group dimension;
  * the linear-attention layers' q/k/v projections stacked to one [384, 128] matrix per layer.
This is one-off host-side preparation (torch ops on the weights are constant folding, not hot path).
"""
import math

import torch

from .synthetic import DINO_PREFIX, DUSTBIN_KEY, EXTRACTOR_PREFIX, HEADS

PATCH_K = 640  # 588 padded to a multiple of the 64-wide K tile


class DeviceWeights:
    pass


def _arch_from_sd(sd, prefix):
    D = sd[prefix + "cls_token"].shape[-1]
    depth = 1 + max(int(k[len(prefix) + 7:].split(".")[0]) for k in sd if k.startswith(prefix + "blocks."))
    return D, depth, D // 64


def fold_basic_block(sd, p):
    """-> (W1 [Cout, 9*Cin], b1, W2 [Cout, 9*Cout (+Cin)], b2, has_shortcut) in fp32."""
    def fold(conv, bn):
        w = sd[p + conv + ".weight"].float()
        scale = sd[p + bn + ".weight"].float() / torch.sqrt(sd[p + bn + ".running_var"].float() + 1e-5)
        bias = sd[p + bn + ".bias"].float() - sd[p + bn + ".running_mean"].float() * scale
        w = (w * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(w.shape[0], -1)  # [Cout, (ky,kx,ci)]
        return w, bias
    w1, b1 = fold("conv1", "bn1")
    w2, b2 = fold("conv2", "bn2")
    sc = sd.get(p + "shortcut.0.weight")
    if sc is not None:
        w2 = torch.cat([w2, sc.float().reshape(sc.shape[0], -1)], 1)
    return w1, b1, w2, b2, sc is not None


def fold_layernorm(w, b, ln_w, ln_b, lp_dtype):
    """LayerNorm folded into the linear layer that consumes it (mickey_hip.h, mk_gemm_ln):
        LN(x) @ W^T + b  =  rstd * (x @ (W diag(ln_w))^T) - rstd * mean * colsum + (b + W @ ln_b)
    -> (W' = W diag(ln_w) rounded to the operand type, colsum = row sums of the ROUNDED W' in fp32, b' in fp32)."""
    w = w.float()
    wf = (w * ln_w.float()[None, :]).to(lp_dtype)
    return wf, wf.float().sum(1), b.float() + w @ ln_b.float()


def prepare_encoder(sd, device, lp_dtype=torch.bfloat16, prefix=DINO_PREFIX, W=None, ln_fold=True, ln_centre=True):
    """ln_fold: also prepare norm1 / norm2 folded into qkv / fc1 (16-bit operand types only): the encoder then runs without
    stand-alone LayerNorm passes (pipeline.encoder_forward)."""
    W = DeviceWeights() if W is None else W
    W.lp = lp_dtype
    dev = device

    def lp(t):
        return t.to(device=dev, dtype=lp_dtype).contiguous()

    def f32(t):
        return t.to(device=dev, dtype=torch.float32).contiguous()

    p = prefix
    D, depth, heads = _arch_from_sd(sd, p)
    W.D, W.depth, W.heads = D, depth, heads
    W.ln_fold = bool(ln_fold) and lp_dtype != torch.float32 and D % 64 == 0
    W.ln_centre = bool(ln_centre)   # row centring of the split residual stream (pipeline.encoder_forward)
    wp = torch.zeros((D, PATCH_K))
    wp[:, :588] = sd[p + "patch_embed.proj.weight"].float().reshape(D, 588)
    W.patch_w, W.patch_b = lp(wp), f32(sd[p + "patch_embed.proj.bias"])
    W.cls = f32(sd[p + "cls_token"].reshape(D))
    W.pos_embed = sd[p + "pos_embed"].float().cpu()  # resampled per grid on demand (host, cached)
    W.blocks = []
    for i in range(depth):
        q = p + "blocks.%d." % i
        blk = DeviceWeights()
        blk.n1w, blk.n1b = f32(sd[q + "norm1.weight"]), f32(sd[q + "norm1.bias"])
        blk.qkv_w, blk.qkv_b = lp(sd[q + "attn.qkv.weight"]), f32(sd[q + "attn.qkv.bias"])
        blk.proj_w, blk.proj_b = lp(sd[q + "attn.proj.weight"]), f32(sd[q + "attn.proj.bias"])
        blk.g1 = f32(sd[q + "ls1.gamma"])
        blk.n2w, blk.n2b = f32(sd[q + "norm2.weight"]), f32(sd[q + "norm2.bias"])
        blk.fc1_w, blk.fc1_b = lp(sd[q + "mlp.fc1.weight"]), f32(sd[q + "mlp.fc1.bias"])
        blk.fc2_w, blk.fc2_b = lp(sd[q + "mlp.fc2.weight"]), f32(sd[q + "mlp.fc2.bias"])
        blk.g2 = f32(sd[q + "ls2.gamma"])
        if W.ln_fold:
            wf, cs, bf = fold_layernorm(sd[q + "attn.qkv.weight"], sd[q + "attn.qkv.bias"], sd[q + "norm1.weight"],
                                        sd[q + "norm1.bias"], lp_dtype)
            blk.qkv_wf, blk.qkv_cs, blk.qkv_bf = lp(wf), f32(cs), f32(bf)
            wf, cs, bf = fold_layernorm(sd[q + "mlp.fc1.weight"], sd[q + "mlp.fc1.bias"], sd[q + "norm2.weight"],
                                        sd[q + "norm2.bias"], lp_dtype)
            blk.fc1_wf, blk.fc1_cs, blk.fc1_bf = lp(wf), f32(cs), f32(bf)
        W.blocks.append(blk)
    W.norm_w, W.norm_b = f32(sd[p + "norm.weight"]), f32(sd[p + "norm.bias"])
    W._pos_cache = {}
    W._pe_cache = {}
    return W


def split_weight_scale(w, target=1024.0):
    """Power-of-two scale of a weight tensor's fp16 planes: `target` (2^10: keeps the lo plane of ordinary weights out of
    fp16's subnormals), lowered for tensors with entries so large (BatchNorm folded over a tiny running variance) that
    w * target would leave fp16's range."""
    m = float(w.abs().max()) if w.numel() else 0.0
    s = float(target)
    while m * s > 32768.0 and s > 2.0 ** -14:
        s *= 0.5
    return s


def split_conv_weight(w, scale=None):
    """[.., Cout, K] fp32 (BatchNorm folded) -> fp16 [.., Cout, 2 K]: the (hi, lo) planes of w * scale (w * scale = hi + lo)
    INTERLEAVED per block of 32 K columns -- 32 hi values, then the 32 lo values of the same columns: one 128-byte LDS row of
    a K step of mk_conv3x3_split / mk_gemm_grouped_split (mickey_hip.h: the four planes a_hi, a_lo, w_hi, w_lo are staged once
    per K step and the products hi.hi + lo.hi + hi.lo are issued from those fragments).  K % 32 == 0.
    scale: a power of two, default split_weight_scale(w); the caller passes the same value as `w_scale` to the kernel wrapper."""
    scale = split_weight_scale(w) if scale is None else scale
    K = w.shape[-1]
    assert K % 32 == 0, "split-operand weights: K must be a multiple of 32"
    ws = w.float() * scale
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    lead = tuple(w.shape[:-1])
    out = torch.stack([hi.reshape(lead + (K // 32, 32)), lo.reshape(lead + (K // 32, 32))], -2).reshape(lead + (2 * K,))
    assert bool(torch.isfinite(out).all()), "weights do not fit fp16 planes"
    return out


def split_conv_weight_planes(planes):
    """Inverse view of split_conv_weight: [.., 2 K] interleaved -> (hi [.., K], lo [.., K]) (tests, diagnostics)."""
    K2 = planes.shape[-1]
    v = planes.reshape(tuple(planes.shape[:-1]) + (K2 // 64, 2, 32))
    lead = tuple(planes.shape[:-1])
    return v[..., 0, :].reshape(lead + (K2 // 2,)), v[..., 1, :].reshape(lead + (K2 // 2,))


def prepare(sd, cfg, device, lp_dtype=torch.bfloat16, heads_dtype=None, ln_fold=True, ln_centre=True, heads_split=False,
            features_lp=False):
    """heads_dtype: operand type of the four head stacks (None = lp_dtype; torch.float32 = as the reference, which always
    runs them in fp32, mickey_extractor.py:53-56) while the encoder uses lp_dtype.  heads_split (with heads_dtype fp32): the
    3x3 convolutions -- 99 % of the heads' flops -- run on the 16-bit matrix cores with split fp16 operands
    (mk_conv3x3_split: fp32-grade products, three MFMA sets on operands staged once), everything else of the heads stays on the
    fp32 path."""
    W = prepare_encoder(sd, device, lp_dtype, ln_fold=ln_fold, ln_centre=ln_centre)
    dev = device
    W.lp_heads = lp_dtype if heads_dtype is None else heads_dtype
    W.heads_split = bool(heads_split)
    W.features_lp = bool(features_lp) and W.heads_split   # the heads' input rounded to fp16 (pipeline.encoder_forward)
    assert not W.heads_split or W.lp_heads == torch.float32, "split-operand convs sit in the fp32 head pipeline"


    def convw(t):   # weights of a 3x3 conv (+ shortcut columns) / a head linear: operand type of the heads, or the split K layout
        if W.heads_split:
            sc = split_weight_scale(t)
            out = split_conv_weight(t, sc).to(device=dev).contiguous()
            out.mk_scale = sc   # the power-of-two scale of its planes travels WITH the tensor object (a copy / .to() of it has
            #                     no such attribute: pipeline.wsc raises instead of reading a stale scale by address)
            return out
        return t.to(device=dev, dtype=W.lp_heads).contiguous()

    def lp(t):
        return t.to(device=dev, dtype=W.lp_heads).contiguous()

    def f32(t):
        return t.to(device=dev, dtype=torch.float32).contiguous()

    # ---- heads: groups in the order det_head, det_offset, depth_head, dsc_head ----
    e = EXTRACTOR_PREFIX
    W.rb = []   # resblock1..3: dict(w1 [4,Co,9Ci], b1 [4,Co], w2 [4,Co,9Co+Ci], b2, cin, cout)
    for b in (1, 2, 3):
        folded = [fold_basic_block(sd, e + h + ".resblock%d." % b) for h in HEADS]
        assert all(f[4] for f in folded), "resblock1-3 change the channel count, so all have a 1x1 shortcut"
        blk = DeviceWeights()
        blk.w1 = convw(torch.stack([f[0] for f in folded]))
        blk.b1 = f32(torch.stack([f[1] for f in folded]))
        blk.w2 = convw(torch.stack([f[2] for f in folded]))
        blk.b2 = f32(torch.stack([f[3] for f in folded]))
        blk.cout = blk.w1.shape[1]
        blk.cin = folded[0][0].shape[1] // 9
        W.rb.append(blk)
    # resblock4: three keypoint heads (128 -> 64, shortcut) and the descriptor head (128 -> 128, identity)
    kp = [fold_basic_block(sd, e + h + ".resblock4.") for h in HEADS[:3]]
    W.rb4_kp = DeviceWeights()
    W.rb4_kp.w1, W.rb4_kp.b1 = convw(torch.stack([f[0] for f in kp])), f32(torch.stack([f[1] for f in kp]))
    W.rb4_kp.w2, W.rb4_kp.b2 = convw(torch.stack([f[2] for f in kp])), f32(torch.stack([f[3] for f in kp]))
    W.rb4_kp.cout = W.rb4_kp.w1.shape[1]
    W.rb4_kp.has_sc = kp[0][4]
    ds = fold_basic_block(sd, e + "dsc_head.resblock4.")
    W.rb4_dsc = DeviceWeights()
    w2d, has_sc = ds[2], ds[4]
    if W.heads_split and not has_sc:   # identity shortcut as identity columns: the split sweeps then carry the block input too
        w2d, has_sc = torch.cat([w2d, torch.eye(w2d.shape[0])], 1), True
    W.rb4_dsc.w1, W.rb4_dsc.b1, W.rb4_dsc.w2, W.rb4_dsc.b2 = convw(ds[0]), f32(ds[1]), convw(w2d), f32(ds[3])
    W.rb4_dsc.cout = ds[0].shape[0]
    W.rb4_dsc.has_sc = has_sc
    # linear-attention stacks
    W.att = []
    for l in range(3):
        lay = DeviceWeights()
        def stk(name):
            return torch.stack([sd[e + h + ".att_layer.layers.%d.%s" % (l, name)].float() for h in HEADS])
        # (split mode: the same interleaved (hi | lo) K layout as the convs, for mk_gemm_grouped_split)
        lay.qkv_w = convw(torch.cat([stk("q_proj.weight"), stk("k_proj.weight"), stk("v_proj.weight")], 1))  # [4,384,128]
        lay.merge_w = convw(stk("merge.weight"))
        lay.mlp0_w, lay.mlp2_w = convw(stk("mlp.0.weight")), convw(stk("mlp.2.weight"))
        lay.n1w, lay.n1b = f32(stk("norm1.weight")), f32(stk("norm1.bias"))
        lay.n2w, lay.n2b = f32(stk("norm2.weight")), f32(stk("norm2.bias"))
        W.att.append(lay)
    W.w_score = f32(sd[e + "det_head.score.weight"].reshape(-1))
    W.w_xy = f32(sd[e + "det_offset.xy_offset.weight"].reshape(2, -1))
    W.w_depth = f32(sd[e + "depth_head.depth.weight"].reshape(-1))
    d = sd.get(DUSTBIN_KEY)
    W.dustbin = float(d) if d is not None else None
    return W


def interp_pos_embed(W, gh, gw, device):
    """Per-resolution constant: bicubic resample of the learned position table (reference
    DINO_modules/dinov2.py:165-189; scale_factor form with the +0.1 fudge, fp32).  Host-side, cached."""
    key = (gh, gw)
    if key not in W._pos_cache:
        pos = W.pos_embed
        n_src = pos.shape[1] - 1
        g = int(math.sqrt(n_src))
        if gh * gw == n_src and gh == gw:
            out = pos[0]
        else:
            table = pos[:, 1:].reshape(1, g, g, -1).permute(0, 3, 1, 2)
            table = torch.nn.functional.interpolate(table, scale_factor=((gh + 0.1) / g, (gw + 0.1) / g), mode="bicubic")
            assert table.shape[-2] == gh and table.shape[-1] == gw
            out = torch.cat([pos[0, :1], table.permute(0, 2, 3, 1).reshape(gh * gw, -1)], 0)
        W._pos_cache[key] = out.to(device=device, dtype=torch.float32).contiguous()
    return W._pos_cache[key]


def sine_pos_table(W, C, h, w, device):
    """[h*w, C] fp32 table of the heads' 2-D sinusoidal encoding (reference att_layers/transformer.py:26-36;
    1-based cell indices).  Constant per grid, cached."""
    key = (C, h, w)
    if key not in W._pe_cache:
        pe = torch.zeros(C, h, w)
        ypos = torch.arange(1, h + 1, dtype=torch.float32).view(1, h, 1).expand(1, h, w)
        xpos = torch.arange(1, w + 1, dtype=torch.float32).view(1, 1, w).expand(1, h, w)
        div = torch.exp(torch.arange(0, C // 2, 2).float() * (-math.log(10000.0) / (C // 2)))[:, None, None]
        pe[0::4] = torch.sin(xpos * div)
        pe[1::4] = torch.cos(xpos * div)
        pe[2::4] = torch.sin(ypos * div)
        pe[3::4] = torch.cos(ypos * div)
        W._pe_cache[key] = pe.reshape(C, h * w).t().contiguous().to(device)
    return W._pe_cache[key]
