"""CPU oracle for the MicKey inference hot path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product path (``mickey_amd``) never does and
fails loudly when its HIP library is missing.

What this is: a functional, fp32, torch-CPU restatement of the reference's algorithm
(nianticlabs/mickey @ 2024-12-23), one plain function per reference function, each citing the
reference file:line it follows.  Weights come in as a flat ``state_dict`` with the reference's
checkpoint key names.

Parity pin: the reference's own tests hold no vectors for this path (its only test file is
benchmark/test_metrics.py, numpy metrics).  The oracle is therefore pinned against OUTPUTS OF THE
REFERENCE ITSELF: ``oracle/make_golden.py`` imports the reference from /root/reference in the build
container, runs it stage by stage on seeded weights/inputs, asserts this restatement agrees
(fp32 round-off for float stages, bit-exact for index stages), and commits the reference's outputs
as fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks the oracle against
those fixtures on every box.
"""
import math

import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------
# DINOv2 ViT/14 encoder


def interp_pos_embed(pos_embed, gh, gw):
    """Bicubic resample of the learned [1, 1+g*g, D] position table to a gh x gw patch grid.
    reference DINO_modules/dinov2.py:165-189 (scale_factor form with the +0.1 fudge; computed in
    fp32).  gh = grid rows (H//14), gw = grid cols (W//14)."""
    pos = pos_embed.float()
    n_src = pos.shape[1] - 1
    g = int(math.sqrt(n_src))
    if gh * gw == n_src and gh == gw:
        return pos
    cls_pos = pos[:, :1]
    table = pos[:, 1:].reshape(1, g, g, -1).permute(0, 3, 1, 2)
    table = F.interpolate(table, scale_factor=((gh + 0.1) / g, (gw + 0.1) / g), mode="bicubic")
    assert table.shape[-2] == gh and table.shape[-1] == gw
    table = table.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)
    return torch.cat([cls_pos, table], 1)


def vit_block(sd, p, x, heads):
    """One pre-LN block with LayerScale: reference DINO_modules/layers/block.py:82-107 (eval branch
    105-106), attention.py:49-62, mlp.py:35-41, layer_scale.py:27-28; LN eps 1e-6 (dinov2.py:87)."""
    B, N, D = x.shape
    dh = D // heads
    y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * dh ** -0.5, qkv[1], qkv[2]
    att = torch.softmax(q @ k.transpose(-2, -1), dim=-1)
    y = (att @ v).transpose(1, 2).reshape(B, N, D)
    y = F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    x = x + y * sd[p + "ls1.gamma"]
    y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    y = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    y = F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + y * sd[p + "ls2.gamma"]


def vit_prepare_tokens(sd, p, img):
    """Patchify (conv k=s=14 == GEMM), prepend CLS, add resampled pos-embed.
    reference dinov2.py:191-200, layers/patch_embed.py:69-82."""
    B, _, H, W = img.shape
    gh, gw = H // 14, W // 14
    tok = F.conv2d(img, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=14)
    tok = tok.flatten(2).transpose(1, 2)
    x = torch.cat([sd[p + "cls_token"].expand(B, -1, -1), tok], 1)
    return x + interp_pos_embed(sd[p + "pos_embed"], gh, gw)


def vit_forward_features(sd, p, img, heads, depth=None, return_tokens=False):
    """x_norm_patchtokens [B, gh*gw, D].  reference dinov2.py:221-236."""
    if depth is None:
        depth = 1 + max(int(k[len(p) + 7:].split(".")[0]) for k in sd if k.startswith(p + "blocks."))
    x = vit_prepare_tokens(sd, p, img)
    for i in range(depth):
        x = vit_block(sd, p + "blocks.%d." % i, x, heads)
    D = x.shape[-1]
    xn = F.layer_norm(x, (D,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)
    if return_tokens:
        return xn[:, 1:], x
    return xn[:, 1:]


# ---------------------------------------------------------------------------------------------
# heads


def basic_block(sd, p, x, relu=True):
    """conv-BN-ReLU-conv-BN-add(shortcut)-ReLU, BN in eval mode (eps 1e-5).
    reference utils/extractor_utils.py:28-35 (+ :18-26 for the layers)."""
    def bn(t, q):
        return F.batch_norm(t, sd[q + "running_mean"], sd[q + "running_var"], sd[q + "weight"], sd[q + "bias"],
                            False, 0.0, 1e-5)
    sc = F.conv2d(x, sd[p + "shortcut.0.weight"]) if (p + "shortcut.0.weight") in sd else x
    y = F.relu(bn(F.conv2d(x, sd[p + "conv1.weight"], padding=1), p + "bn1."))
    y = bn(F.conv2d(y, sd[p + "conv2.weight"], padding=1), p + "bn2.") + sc
    return F.relu(y) if relu else y


def sine_pos_encoding(d_model, h, w):
    """2-D sinusoidal table [d_model, h, w]; 1-based cell indices.
    reference att_layers/transformer.py:26-36."""
    pe = torch.zeros(d_model, h, w)
    ypos = torch.arange(1, h + 1, dtype=torch.float32).view(1, h, 1).expand(1, h, w)
    xpos = torch.arange(1, w + 1, dtype=torch.float32).view(1, 1, w).expand(1, h, w)
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))[:, None, None]
    pe[0::4] = torch.sin(xpos * div)
    pe[1::4] = torch.cos(xpos * div)
    pe[2::4] = torch.sin(ypos * div)
    pe[3::4] = torch.cos(ypos * div)
    return pe


def linear_attention(q, k, v, eps=1e-6):
    """'Transformers are RNNs' linear attention, phi = elu+1.  q,k,v [B, L, H, d].
    reference att_layers/attention.py:46-64."""
    Q = F.elu(q) + 1.0
    K = F.elu(k) + 1.0
    L = v.shape[1]
    v = v / L
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1.0 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + eps)
    return torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * L


def encoder_layer(sd, p, x, nhead=8):
    """LoFTR-style encoder layer (self-attention: source == x).
    reference att_layers/transformer_utils.py:40-66; LN eps 1e-5 (nn.LayerNorm default)."""
    B, L, C = x.shape
    d = C // nhead
    q = F.linear(x, sd[p + "q_proj.weight"]).view(B, L, nhead, d)
    k = F.linear(x, sd[p + "k_proj.weight"]).view(B, L, nhead, d)
    v = F.linear(x, sd[p + "v_proj.weight"]).view(B, L, nhead, d)
    msg = linear_attention(q, k, v).reshape(B, L, C)
    msg = F.linear(msg, sd[p + "merge.weight"])
    msg = F.layer_norm(msg, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    msg = F.linear(F.relu(F.linear(torch.cat([x, msg], 2), sd[p + "mlp.0.weight"])), sd[p + "mlp.2.weight"])
    msg = F.layer_norm(msg, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    return x + msg


def self_att_stack(sd, p, feats, add_pos_enc, num_layers=3):
    """reference att_layers/transformer.py:75-103."""
    B, C, h, w = feats.shape
    if add_pos_enc:
        feats = feats + sine_pos_encoding(C, h, w)[None]
    x = feats.flatten(2).transpose(1, 2)
    for l in range(num_layers):
        x = encoder_layer(sd, p + "layers.%d." % l, x)
    return x.transpose(1, 2).reshape(B, C, h, w)


def head_trunk(sd, hp, feat, add_pos_enc, last_relu=True):
    """resblock1-3 -> att_layer -> resblock4 (reference mickey_extractor.py:126-132 and siblings)."""
    x = basic_block(sd, hp + "resblock1.", feat)
    x = basic_block(sd, hp + "resblock2.", x)
    x = basic_block(sd, hp + "resblock3.", x)
    x = self_att_stack(sd, hp + "att_layer.", x, add_pos_enc)
    return basic_block(sd, hp + "resblock4.", x, relu=last_relu)


def border_softmax(scores, border=3, temperature=100.0, eps=1e-16):
    """Mean-subtracted, temperature-100 spatial softmax with a zeroed border.
    reference mickey_extractor.py:98-124."""
    B = scores.shape[0]
    s = scores - (scores.reshape(B, -1).mean(-1).view(B, 1, 1, 1) + eps)
    e = torch.exp(s / temperature)
    mask = torch.zeros_like(e)
    mask[:, :, border:e.shape[2] - border, border:e.shape[3] - border] = 1.0
    e = e * mask
    return e / (e.sum((-1, -2), keepdim=True) + eps)


def extractor_heads(sd, cfg, feat, prefix="compute_matches.extractor."):
    """Four heads on the [B,C,h,w] fp32 feature volume -> (kpts, depths, scrs, dscs).
    reference mickey_extractor.py:53-58, 126-140, 166-178, 202-218, 237-251."""
    mk = cfg["MICKEY"]
    kp_pe = mk["KP_HEADS"]["POS_ENCODING"]
    x = head_trunk(sd, prefix + "det_head.", feat, kp_pe)
    s = F.conv2d(x, sd[prefix + "det_head.score.weight"])
    if mk["KP_HEADS"]["USE_SOFTMAX"]:
        scrs = border_softmax(s, 3)
    else:
        m = torch.zeros_like(s)
        m[:, :, 3:-3, 3:-3] = 1.0
        scrs = torch.sigmoid(s) * m
    x = head_trunk(sd, prefix + "det_offset.", feat, kp_pe)
    kpts = torch.sigmoid(F.conv2d(x, sd[prefix + "det_offset.xy_offset.weight"]))
    x = head_trunk(sd, prefix + "depth_head.", feat, kp_pe)
    depths = F.conv2d(x, sd[prefix + "depth_head.depth.weight"])
    if mk["KP_HEADS"]["USE_DEPTHSIGMOID"]:
        depths = mk["KP_HEADS"]["MAX_DEPTH"] * torch.sigmoid(depths)
    x = head_trunk(sd, prefix + "dsc_head.", feat, mk["DSC_HEAD"]["POS_ENCODING"], last_relu=False)
    if mk["DSC_HEAD"]["NORM_DSC"]:
        x = x / x.pow(2).sum(1, keepdim=True).add(1e-10).pow(0.5)  # extractor_utils.py:6-10
    return kpts, depths, scrs, x


def extractor_forward(sd, cfg, img, heads=16, prefix="compute_matches.extractor.", return_feat=False):
    """Crop to /14, encoder, heads.  reference mickey_extractor.py:43-58.  fp32 throughout
    (FLOAT16=False gold)."""
    B, _, H, W = img.shape
    f = cfg["MICKEY"]["DINOV2"]["DOWN_FACTOR"]
    img = img[:, :, : f * (H // f), : f * (W // f)]
    tok = vit_forward_features(sd, prefix + "dinov2_vitl14.", img, heads)
    feat = tok.permute(0, 2, 1).reshape(B, tok.shape[-1], H // f, W // f).float()
    out = extractor_heads(sd, cfg, feat, prefix)
    return out + (feat,) if return_feat else out


# ---------------------------------------------------------------------------------------------
# correspondences / matcher


def abs_keypoints(kpts, down):
    """(sigmoid offset + cell index) * down; channel 0 = x (col), 1 = y (row).
    reference compute_correspondences.py:20-31."""
    B, _, h, w = kpts.shape
    xs = torch.arange(w, dtype=kpts.dtype).view(1, 1, 1, w).expand(B, 1, h, w)
    ys = torch.arange(h, dtype=kpts.dtype).view(1, 1, h, 1).expand(B, 1, h, w)
    return (kpts + torch.cat([xs, ys], 1)) * down


def dual_softmax(dsc0, dsc1, dustbin, temperature=0.1):
    """reference utils/feature_matcher.py:64-83 (dustbin branch when ``dustbin`` is not None)."""
    S = torch.matmul(dsc0.transpose(1, 2).contiguous(), dsc1) / temperature
    if dustbin is None:
        return F.softmax(S, 1) * F.softmax(S, 2)
    b, m, n = S.shape
    d = torch.as_tensor(dustbin, dtype=S.dtype)
    Sp = torch.cat([torch.cat([S, d.expand(b, m, 1)], -1), torch.cat([d.expand(b, 1, n), d.expand(b, 1, 1)], -1)], 1)
    P = F.softmax(Sp, 1) * F.softmax(Sp, 2)
    return P[:, :-1, :-1]


def sinkhorn(dsc0, dsc1, alpha, iters=10, descriptor_dim=128):
    """Log-domain optimal transport with dustbins.
    reference utils/feature_matcher.py:93-137 (forward :125-137 is unreachable through
    featureMatcher.forward -- SURVEY D4 -- the maths is what is restated)."""
    S = torch.einsum("bdn,bdm->bnm", dsc0, dsc1) / descriptor_dim ** 0.5
    b, m, n = S.shape
    a = torch.as_tensor(alpha, dtype=S.dtype)
    Z = torch.cat([torch.cat([S, a.expand(b, m, 1)], -1), torch.cat([a.expand(b, 1, n), a.expand(b, 1, 1)], -1)], 1)
    ms, ns = torch.tensor(float(m)), torch.tensor(float(n))
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])[None].expand(b, -1)
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])[None].expand(b, -1)
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return (Z + u.unsqueeze(2) + v.unsqueeze(1) - norm).exp()[:, :-1, :-1]


def mutual_nn_matches(scores):
    """Mutual nearest neighbours on scores[:, :-1, :-1], B == 1, sorted by score (descending).
    Returns int64 [m, 2].  reference utils/feature_matcher.py:19-46 (min_conf = 0)."""
    sub = scores[:, :-1, :-1]
    max0, max1 = sub.max(2), sub.max(1)
    i0, i1 = max0.indices, max1.indices
    rows = torch.arange(i0.shape[1])[None]
    mutual = rows == i1.gather(1, i0)
    valid = mutual & (max0.values.exp() > 0.0)
    idx0 = rows[0][valid[0]]
    idx1 = i0[0][valid[0]]
    sc = scores[0, idx0, idx1]
    order = torch.sort(sc, descending=True).indices
    return torch.stack([idx0, idx1], 1)[order]


def compute_correspondences(sd, cfg, data, heads=16):
    """reference compute_correspondences.py:52-92 plus compute_pose.py:23 (final_scores)."""
    out = {}
    f = cfg["MICKEY"]["DINOV2"]["DOWN_FACTOR"]
    per_img = []
    for key in ("image0", "image1"):
        kpts, depths, scrs, dscs = extractor_forward(sd, cfg, data[key], heads)
        per_img.append((abs_keypoints(kpts, f), depths, scrs, dscs))
    for i, (kp, dp, sc, ds) in enumerate(per_img):
        B, _, h, w = kp.shape
        out["kps%d_shape" % i] = [h, w]
        out["depth%d_map" % i] = dp
        out["kps%d" % i] = kp.reshape(B, 2, h * w)
        out["depth_kp%d" % i] = dp.reshape(B, 1, h * w)
        out["scr%d" % i] = sc.reshape(B, 1, h * w)
        out["dsc%d" % i] = ds.reshape(B, ds.shape[1], h * w)
    fm = cfg["FEATURE_MATCHER"]
    dustbin = sd.get("compute_matches.matcher.matching_mat.dustbin_score")
    if fm["TYPE"] == "DualSoftmax":
        out["scores"] = dual_softmax(out["dsc0"], out["dsc1"],
                                     dustbin if fm["DUAL_SOFTMAX"]["USE_DUSTBIN"] else None,
                                     fm["DUAL_SOFTMAX"]["TEMPERATURE"])
    else:
        out["scores"] = sinkhorn(out["dsc0"], out["dsc1"], dustbin, fm["SINKHORN"]["NUM_IT"])
    out["kp_scores"] = torch.matmul(out["scr0"].transpose(2, 1).contiguous(), out["scr1"])  # :46-50
    out["final_scores"] = out["scores"] * out["kp_scores"]
    out["down_factor"] = f
    return out


# ---------------------------------------------------------------------------------------------
# probabilistic Procrustes solver


def exp_race_topk(p, k, noise=None, generator=None):
    """Weighted sampling without replacement == top-k of p / Exp(1) noise; identical, draw for draw,
    to ``torch.multinomial(p, k)`` on CPU (that is how ATen implements it; SURVEY §8(c) probe).
    Returns (indices [rows,k] int64 sorted by key descending, noise)."""
    # torch.multinomial's input checks (they are what sends the reference into its zero-pose branch)
    if not bool(torch.isfinite(p).all()) or bool((p < 0).any()):
        raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0")
    if bool((p.sum(-1) <= 0).any()):
        raise RuntimeError("invalid multinomial distribution (sum of probabilities <= 0)")
    if noise is None:
        noise = torch.empty_like(p).exponential_(1.0, generator=generator)
    return torch.topk(p / noise, k, dim=-1).indices, noise


def backproject(uv, depth, K):
    """X = depth * K^-1 [u, v, 1]^T.  uv [M,k,2], depth [M,k,1], K [M,3,3].
    reference utils/training_utils.py:7-22."""
    M, k, _ = uv.shape
    uv1 = torch.cat([uv, torch.ones(M, k, 1)], -1)
    return depth * (torch.linalg.inv(K) @ uv1.transpose(2, 1)).transpose(2, 1)


def kabsch(A, Bp, w=None, masked=False, eps=1e-16):
    """Rigid fit B ~ A R^T + t.  Unweighted branch (w None): reference loss/solvers.py:31-39;
    masked-weighted branch: :13-26 (centroids with w/(sum|w|+eps), covariance with RAW w);
    SVD + reflection fix + R = V Z U^T, t = b_mean - a_mean R^T: :45-52."""
    if w is None:
        a_mean = A.mean(1, keepdim=True)
        b_mean = Bp.mean(1, keepdim=True)
        H = (A - a_mean).transpose(1, 2) @ (Bp - b_mean)
    else:
        wn = (w / (w.abs().sum(1, keepdim=True) + eps)).unsqueeze(-1)
        a_mean = (wn * A).sum(1, keepdim=True)
        b_mean = (wn * Bp).sum(1, keepdim=True)
        ww = w.unsqueeze(-1) if masked else wn
        H = (A - a_mean).transpose(1, 2) @ (ww * (Bp - b_mean))
    U, S, V = torch.svd(H)
    Z = torch.eye(3).repeat(A.shape[0], 1, 1)
    Z[:, 2, 2] = torch.sign(torch.linalg.det(U @ V.transpose(1, 2)))
    R = V @ Z @ U.transpose(1, 2)
    t = b_mean - a_mean @ R.transpose(1, 2)
    return R, t, H


def point_dist(X, Y, R, t):
    """sqrt(|R X + t - Y|^2 + 1e-6).  reference utils/training_utils.py:58-59 / 72-73."""
    Xt = (R @ X.transpose(2, 1)).transpose(2, 1) + t
    return ((Xt - Y).pow(2).sum(-1) + 1e-6).pow(0.5)


def soft_inliers(X, Y, R, t, th):
    """reference utils/training_utils.py:55-61."""
    return torch.sigmoid((5.0 / th) * (th - point_dist(X, Y, R, t))).sum(-1, keepdim=True)


def hard_inliers(X, Y, R, t, th):
    """reference utils/training_utils.py:71-75."""
    return ((th - point_dist(X, Y, R, t)) >= 0).float()


def refine_pose(Xb, Yb, R, t, th, num_ref, k_min):
    """<= num_ref rounds of {hard-inlier recount -> masked Kabsch}.  Per-pair rule: refine iff
    count >= k_min and count > best so far (initialised to k_min); stop when no pair refines.
    reference probabilisticProcrustes.py:283-300."""
    R, t = R.clone(), t.clone()
    B = Xb.shape[0]
    mask_ref = torch.zeros(B, Xb.shape[1])
    best = torch.full((B,), float(k_min))
    rounds = 0
    for _ in range(num_ref):
        inl = hard_inliers(Xb, Yb, R, t, th)
        cnt = inl.sum(-1)
        do = (cnt >= k_min) & (cnt > best)
        best[do] = cnt[do]
        if int(do.sum()) == 0:
            break
        rounds += 1
        mask_ref[do] = inl[do]
        Rn, tn, _ = kabsch(Xb[do], Yb[do], mask_ref[do], masked=True)
        R[do], t[do] = Rn, tn
    return R, t, rounds


def estimate_pose(data, cfg, return_inliers=False, noise_outer=None, noise_inner=None, generator=None,
                  return_debug=False):
    """Vectorised probabilistic-Procrustes RANSAC: reference probabilisticProcrustes.py:183-348.
    ``noise_outer`` [B*IT_MATCHES, n*n] / ``noise_inner`` [B*IT_MATCHES*IT_RANSAC, NUM_SAMPLED]
    inject the Exp(1) draws; when None they are drawn from torch's generator in the same order
    the reference's two ``torch.multinomial`` calls consume it, so with equal seeds the output is
    bit-identical to the reference on CPU."""
    P = cfg["PROCRUSTES"]
    it_m, it_r, ns, k3 = P["IT_MATCHES"], P["IT_RANSAC"], P["NUM_SAMPLED_MATCHES"], P["NUM_CORR_3D_3D"]
    fs = data["final_scores"]
    B, n, _ = fs.shape
    kps0, kps1, d0, d1 = data["kps0"], data["kps1"], data["depth_kp0"], data["depth_kp1"]
    K0, K1 = data["K_color0"].float(), data["K_color1"].float()
    dbg = {}
    zero = (torch.zeros(B, 3, 3), torch.zeros(B, 1, 3), torch.zeros(B))
    try:
        rows = fs.reshape(B, 1, n * n).expand(B, it_m, n * n).reshape(B * it_m, n * n)
        idx, noise_outer = exp_race_topk(rows, ns, noise_outer, generator)          # :230-231
        i0 = torch.div(idx, n, rounding_mode="trunc")                                # :233
        i1 = idx % n                                                                 # :234
        bsel = torch.arange(B).repeat_interleave(it_m)[:, None].expand(-1, ns)
        cor0 = kps0[bsel, :, i0]                                                     # [B*it_m, ns, 2]
        cor1 = kps1[bsel, :, i1]
        dd0 = d0[bsel, :, i0]
        dd1 = d1[bsel, :, i1]
        wts = fs.reshape(B, n * n)[bsel, idx]                                        # :241
        X = backproject(cor0, dd0, K0.repeat_interleave(it_m, 0))                    # :243-244
        Y = backproject(cor1, dd1, K1.repeat_interleave(it_m, 0))
        wv = wts.repeat_interleave(it_r, 0)                                          # :249
        idx3, noise_inner = exp_race_topk(wv, k3, noise_inner, generator)           # :251
        gsel = torch.arange(B * it_m).repeat_interleave(it_r)[:, None].expand(-1, k3)
        Xk, Yk = X[gsel, idx3], Y[gsel, idx3]                                        # :254-255
        Rh, th_, Hh = kabsch(Xk, Yk)                                                 # :259
        invalid = bool(torch.isnan(th_).any() or torch.isinf(th_).any() or torch.isnan(Rh).any()
                       or torch.isinf(Rh).any())                                     # :261-262
        Xv, Yv = X.repeat_interleave(it_r, 0), Y.repeat_interleave(it_r, 0)
        score = soft_inliers(Xv, Yv, Rh, th_, P["TH_SOFT_INLIER"]).reshape(B, it_m * it_r)   # :265-268
        best = torch.argmax(score, 1)                                                # :275
        ar = torch.arange(B)
        R = Rh.reshape(B, it_m * it_r, 3, 3)[ar, best]
        t = th_.reshape(B, it_m * it_r, 1, 3)[ar, best]
        outer = best // it_r
        Xb = X.reshape(B, it_m, ns, 3)[ar, outer]
        Yb = Y.reshape(B, it_m, ns, 3)[ar, outer]
        if return_debug:
            dbg.update(idx=idx, idx3=idx3, X=X, Y=Y, weights=wts, R_hyp=Rh, t_hyp=th_, H_hyp=Hh, score=score,
                       best=best, R_best=R.clone(), t_best=t.clone(), X_best=Xb, Y_best=Yb,
                       noise_outer=noise_outer, noise_inner=noise_inner)
        R, t, rounds = refine_pose(Xb, Yb, R, t, P["TH_INLIER"], P["NUM_REFINEMENTS"], k3)      # :283-300
        conf = soft_inliers(Xb, Yb, R, t, P["TH_INLIER"])                            # :303
        inl_list = [torch.zeros(0, 5)] * B
        if return_inliers:                                                           # :306-327
            m = hard_inliers(Xb, Yb, R, t, P["TH_INLIER"])
            c0 = cor0.reshape(B, it_m, ns, 2)[ar, outer]
            c1 = cor1.reshape(B, it_m, ns, 2)[ar, outer]
            e0 = dd0.reshape(B, it_m, ns, 1)[ar, outer]
            e1 = dd1.reshape(B, it_m, ns, 1)[ar, outer]
            ww = wts.reshape(B, it_m, ns)[ar, outer]
            inl_list = []
            for b in range(B):
                sel = m[b] == 1.0
                order = torch.argsort(ww[b, sel], descending=True)
                inl_list.append(torch.cat([c0[b, sel][order], c1[b, sel][order], ww[b, sel][order].unsqueeze(-1),
                                           e0[b, sel][order], e1[b, sel][order]], 1))
        if return_debug:
            dbg.update(rounds=rounds)
        if invalid:
            R, t, conf = zero
            inl_list = [torch.zeros(0, 5)] * B
    except Exception:  # reference :331-336 swallows everything into a zero pose
        print("[Except Reached]: Invalid Procrustes configuration! ")
        R, t, conf = zero
        inl_list = [torch.zeros(0, 5)] * B
    out = (R, t, conf, inl_list) if return_inliers else (R, t, conf)
    return out + (dbg,) if return_debug else out


def mickey_forward(sd, cfg, data, return_inliers=False, heads=16, generator=None):
    """reference compute_pose.py:20-37."""
    data.update(compute_correspondences(sd, cfg, data, heads))
    res = estimate_pose(data, cfg, return_inliers, generator=generator)
    data["R"], data["t"], data["inliers"] = res[0], res[1], res[2]
    if return_inliers:
        data["inliers_list"] = res[3]
    return res[0], res[1]
