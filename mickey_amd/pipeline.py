"""Host-side orchestration of the HIP kernels for the MicKey hot path: which kernel runs on which
buffer, in which order.  All arithmetic is in libmickey_hip.so; torch provides device memory and
the stream only.  Mirrors (file:line under the reference):
  encoder   DINO_modules/dinov2.py:191-236, layers/block.py:105-106
  heads     mickey_extractor.py:53-58,126-140,166-178,202-218,237-251
  matcher   compute_correspondences.py:52-92, compute_pose.py:23
  solver    utils/probabilisticProcrustes.py:183-348
"""
import torch

from . import ops
from . import weights as wts_mod


class Workspace:
    """Per-shape device buffers, allocated once and reused across forward calls."""

    def __init__(self):
        self.bufs = {}
        self.sat_flag = None   # split-operand heads: the int32 [1] device word the plane-writing kernels report saturation into

    def get_planes(self, name, shape, device, zero=False):
        """(hi, lo) fp16 planes of a split-operand activation: the two halves of ONE buffer (the kernels address both planes of
        a source from one base pointer, mickey_hip.h)."""
        t = self.get(name, (2,) + tuple(shape), torch.float16, device, zero=zero)
        return t[0], t[1]

    def get(self, name, shape, dtype, device, zero=False):
        key = (name, tuple(shape), dtype)
        t = self.bufs.get(key)
        if t is None or t.device != device:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=device)
            self.bufs[key] = t
        return t


def encoder_forward(W, ws, img):
    """img fp32 [nimg, 3, H, W] on device (already cropped view or not: the /14 crop is implicit in
    gh, gw), or a sequence of such tensors of one H x W (the two image sets of a pair batch: patched straight into one
    token matrix, no concatenated copy of the images).  Returns the final-norm patch tokens, lp, as a bordered feature map
    of nimg gh x gw grids."""
    imgs = list(img) if isinstance(img, (list, tuple)) else [img]
    img = imgs[0]
    dev, lp = img.device, W.lp
    _, _, H, Wd = img.shape
    assert all(t.shape[1:] == img.shape[1:] for t in imgs)
    nimg = sum(t.shape[0] for t in imgs)
    gh, gw = H // 14, Wd // 14
    npatch, D, heads = gh * gw, W.D, W.heads
    ntok = npatch + 1
    pad = (ntok + 63) // 64 * 64
    M = nimg * ntok
    pos = wts_mod.interp_pos_embed(W, gh, gw, dev)
    a = ws.get("im2col", (nimg * npatch, wts_mod.PATCH_K), lp, dev)
    r0 = 0
    for t in imgs:
        ops.im2col_patch14(t, gh, gw, wts_mod.PATCH_K, lp, out=a[r0:r0 + t.shape[0] * npatch])
        r0 += t.shape[0] * npatch
    x = ws.get("x", (M, D), torch.float32, dev)
    att = ws.get("att", (M, D), lp, dev)
    hid = ws.get("hid", (M, 4 * D), lp, dev)
    q = ws.get("q", (nimg, heads, pad, 64), lp, dev, zero=True)   # pad rows stay zero forever
    k = ws.get("k", (nimg, heads, pad, 64), lp, dev, zero=True)
    vt = ws.get("vt", (nimg, heads, 64, pad), lp, dev, zero=True)
    if getattr(W, "ln_fold", False):
        # norm1 / norm2 folded into the GEMMs around them (mickey_hip.h, mk_gemm_*_ln): the residual stream lives as two
        # 16-bit planes (x = xh + xl); whoever writes it also writes per-slot row statistics; qkv / fc1 read xh (raw) and
        # normalise in their epilogue.  The last block writes fp32 rows for the final norm.
        xh = ws.get("xh", (M, D), lp, dev)
        xl = ws.get("xl", (M, D), lp, dev)
        st = ws.get("ln_stats", (M, D // 64, 2), torch.float32, dev)
        # row centring (mickey_hip.h): every consumer publishes the rows' means, the producer after it takes them off -- the
        # stream stays centred (LayerNorm, its only reader, cannot tell), so the raw hi plane rounds x - mean, not x
        sh = ws.get("ln_shift", (M,), torch.float32, dev) if getattr(W, "ln_centre", True) else None
        ops.gemm_patch_embed_ln(a, W.patch_w, W.patch_b, pos, xh, xl, st, nimg, npatch)
        ops.cls_token_ln(W.cls, pos, xh, xl, st, nimg, ntok, D)
        if sh is not None:
            ops.recentre_split(xh, xl, st)   # the first consumer (qkv of block 0) meets centred rows too
        last = len(W.blocks) - 1
        for bi, blk in enumerate(W.blocks):
            ops.gemm_qkv_ln(xh, blk.qkv_wf, blk.qkv_bf, blk.qkv_cs, st, 1e-6, q, k, vt, nimg, ntok, pad, heads, shift_out=sh)
            ops.flash_attn(q, k, vt, att, nimg, heads, ntok, pad)
            ops.gemm_ls_residual_ln(att, blk.proj_w, blk.proj_b, blk.g1, xh, xl, st, shift=sh)
            ops.gemm_ln(xh, blk.fc1_wf, blk.fc1_bf, blk.fc1_cs, st, 1e-6, act=ops.ACT_GELU, out=hid, shift_out=sh)
            ops.gemm_ls_residual_ln(hid, blk.fc2_w, blk.fc2_b, blk.g2, xh, xl, st, x_out=x if bi == last else None, shift=sh)
    else:
        ops.gemm_patch_embed(a, W.patch_w, W.patch_b, pos, x, nimg, npatch)
        ops.cls_token(W.cls, pos, x, nimg, ntok, D)
        y = ws.get("y", (M, D), lp, dev)
        for blk in W.blocks:
            ops.layernorm(x, blk.n1w, blk.n1b, 1e-6, out=y)
            ops.gemm_qkv(y, blk.qkv_w, blk.qkv_b, q, k, vt, nimg, ntok, pad, heads)
            ops.flash_attn(q, k, vt, att, nimg, heads, ntok, pad)
            ops.gemm_ls_residual(att, blk.proj_w, blk.proj_b, blk.g1, x)
            ops.layernorm(x, blk.n2w, blk.n2b, 1e-6, out=y)
            ops.gemm(y, blk.fc1_w, blk.fc1_b, act=ops.ACT_GELU, out=hid)
            ops.gemm_ls_residual(hid, blk.fc2_w, blk.fc2_b, blk.g2, x)
    # the heads' operand type, as a BORDERED feature map (mickey_hip.h: what a 3x3 conv reads; border rows stay zero)
    # (bordered buffers are zeroed ONCE and only their pixel rows are ever written: the key carries the geometry, because two
    # geometries can share a row count while their border rows sit elsewhere)
    R = ops.bordered_rows(nimg, gh, gw)
    if getattr(W, "heads_split", False):
        # split-operand heads: the final norm writes the convs' (hi, lo) fp16 operand planes directly (no fp32 feature map).
        # features_lp (AMD.FEATURES_LP; automatic behind an fp16 encoder): the reference's heads receive what its fp16 encoder
        # returns -- fp16 values, widened exactly by .float() (mickey_extractor.py:49-52) -- so only the hi plane is written (the rows
        # rounded to fp16) and the lo plane of this buffer, zeroed at allocation, stays zero: the first conv of every head then runs
        # two products instead of three (heads_forward)
        flp = bool(getattr(W, "features_lp", False))
        feat = ws.get_planes("feat_hl%s_%d_%d_%d" % ("_lp" if flp else "", nimg, gh, gw), (R, D), dev, zero=True)
    else:
        feat = ws.get("feat_%d_%d_%d" % (nimg, gh, gw), (R, D), getattr(W, "lp_heads", lp), dev, zero=True)
    hi_only = isinstance(feat, tuple) and bool(getattr(W, "features_lp", False))
    ops.layernorm(x, W.norm_w, W.norm_b, 1e-6, out=(feat[0], None) if hi_only else feat, rows_out=nimg * npatch, rows_per_img=ntok, skip=1,
                  bordered=(nimg, gh, gw), sat=ws.sat_flag)
    return feat, gh, gw


def heads_forward(W, ws, feat, nimg, gh, gw, cfg):
    """feat lp, bordered feature map of nimg gh x gw grids [bordered_rows, D] (split-operand heads: its (hi, lo) fp16 planes)
    -> scr [nimg,1,n], kps [nimg,2,n]
    (absolute pixels), depth [nimg,1,n], dsc [nimg,Cd,n], all fp32.  Every activation a 3x3 conv reads is bordered (zeroed
    once at allocation, the kernels write pixels only); what only row-wise kernels read is dense."""
    dev, lp = (feat[0] if isinstance(feat, tuple) else feat).device, getattr(W, "lp_heads", W.lp)
    n = gh * gw
    M = nimg * n
    G = 4
    mk = cfg["MICKEY"]
    R = ops.bordered_rows(nimg, gh, gw)
    geo = "_%d_%d_%d" % (nimg, gh, gw)   # bordered buffers: the workspace key carries the geometry (see encoder_forward)
    split = bool(getattr(W, "heads_split", False))   # 3x3 convs on split fp16 operands inside the fp32 head pipeline
    sat = ws.sat_flag   # the plane-writing kernels report saturation / NaN here (per workspace = per model: no shared word)

    def wsc(w):   # power-of-two scale of a split weight tensor's planes (set by weights.prepare on the tensor object)
        sc = getattr(w, "mk_scale", None)
        if sc is None:
            raise RuntimeError("split-operand weight without its plane scale: the tensor is not the one weights.prepare made "
                               "(copied / moved weights must be re-prepared)")
        return sc

    def plane_pair(name, shape):
        """Zeroed (hi, lo) fp16 planes of a bordered activation that a split conv WRITES (the next split conv's operands)."""
        return ws.get_planes(name + "_hl" + geo, shape, dev, zero=True)

    x_in, c_in, s_in = feat, W.D, 0   # first block: all four heads read the same feature map
    xp = feat if split else None   # (the final norm wrote the planes)
    for bi, rb in enumerate(W.rb):
        co = rb.cout
        last = bi == len(W.rb) - 1   # its output feeds the attention layers (row-wise kernels): dense rows
        if split:
            # conv -> conv inside the stack: the epilogue writes the next conv's (hi, lo) operand planes directly (no fp32
            # round trip, no mk_split_planes pass); only the stack's output (read by the row-wise attention kernels) is fp32
            hp = plane_pair("rb%d_h" % bi, (G, R, co))
            # (the first block reads the feature map: with features_lp its lo plane is identically zero -- hi only, two products)
            x1 = (xp[0], None) if bi == 0 and getattr(W, "features_lp", False) else xp
            ops.conv3x3_split(x1, c_in, rb.w1, rb.b1, hp, co, G, nimg, gh, gw, act=ops.ACT_RELU, stride_in1=s_in, w_scale=wsc(rb.w1), sat=sat,
                              stride_w=rb.w1.shape[1] * rb.w1.shape[2], stride_bias=co, stride_out=R * co, out_bordered=True)
            xo = ws.get("rb%d_x" % bi + geo, (G, M, co), lp, dev) if last else plane_pair("rb%d_x" % bi, (G, R, co))
            ops.conv3x3_split(hp, co, rb.w2, rb.b2, xo, co, G, nimg, gh, gw, act=ops.ACT_RELU, in2=xp, C2=c_in, w_scale=wsc(rb.w2), sat=sat,
                              stride_in1=R * co, stride_in2=s_in, stride_w=rb.w2.shape[1] * rb.w2.shape[2], stride_bias=co,
                              stride_out=(M if last else R) * co, out_bordered=not last)
            if not last:
                xp = xo
        else:
            h1 = ws.get("rb%d_h" % bi + geo, (G, R, co), lp, dev, zero=True)
            xo = ws.get("rb%d_x" % bi + geo, (G, M if last else R, co), lp, dev, zero=not last)
            ops.conv3x3(x_in, c_in, rb.w1, rb.b1, h1, co, G, nimg, gh, gw, act=ops.ACT_RELU, stride_in1=s_in,
                        stride_w=rb.w1.shape[1] * rb.w1.shape[2], stride_bias=co, stride_out=R * co, out_bordered=True)
            ops.conv3x3(h1, co, rb.w2, rb.b2, xo, co, G, nimg, gh, gw, act=ops.ACT_RELU, in2=x_in, C2=c_in,
                        stride_in1=R * co, stride_in2=s_in, stride_w=rb.w2.shape[1] * rb.w2.shape[2], stride_bias=co,
                        stride_out=(M if last else R) * co, out_bordered=not last)
        x_in, c_in, s_in = xo, co, R * co
    C = c_in  # 128
    # ---- Transformer_self_att: 3 linear-attention encoder layers, residual stream in fp32 ----
    xs = ws.get("att_xs", (G, M, C), torch.float32, dev)
    cat = ws.get("att_cat", (G, M, 2 * C), lp, dev)
    pe = wts_mod.sine_pos_table(W, C, gh, gw, dev)
    kp_pe, dsc_pe = bool(mk["KP_HEADS"]["POS_ENCODING"]), bool(mk["DSC_HEAD"]["POS_ENCODING"])
    if kp_pe == dsc_pe:
        ops.posenc_add(x_in, pe if kp_pe else None, xs, cat, G, nimg, n, C)
    else:
        ops.posenc_add(x_in[:3], pe if kp_pe else None, xs[:3], cat[:3], 3, nimg, n, C)
        ops.posenc_add(x_in[3:], pe if dsc_pe else None, xs[3:], cat[3:], 1, nimg, n, C)
    qkv = ws.get("att_qkv", (G, M, 3 * C), torch.float32, dev)
    kv = ws.get("att_kv", (G * nimg * (C // 16), 272), torch.float32, dev)
    kvw = ws.get("att_kvw", (ops.linattn_work_floats(G, nimg, n, C),), torch.float32, dev)
    msg = ws.get("att_msg", (G, M, C), lp, dev)
    mrg = ws.get("att_mrg", (G, M, C), torch.float32, dev)
    hid = ws.get("att_hid", (G, M, 2 * C), lp, dev)
    if split:
        x4 = None
        x4h, x4l = plane_pair("att_out", (G, R, C))   # read by resblock4's convs: bordered planes, written by the last LayerNorm
    else:
        x4 = ws.get("att_out" + geo, (G, R, C), lp, dev, zero=True)   # read by resblock4's convs: bordered
    nl = len(W.att)
    if split:
        # the layers' small linears on split operands too (fp32 MFMA: 4.9 ms per 32 pairs; split: the conversions below + 12
        # HBM-bound GEMMs): planes of `cat` (its two column halves are rewritten at different times), of `msg`, and `hid`
        # written as planes by the GEMM that produces it
        catp = ws.get_planes("att_cat_hl", (G, M, 2 * C), dev)
        msgp = ws.get_planes("att_msg_hl", (G, M, C), dev)
        hidp = ws.get_planes("att_hid_hl", (G, M, 2 * C), dev)
    for li, lay in enumerate(W.att):
        last = li == nl - 1
        if split:
            if li == 0:   # later layers: the previous layer's closing LayerNorm wrote these planes
                ops.split_planes(cat[:, :, :C], catp[0][:, :, :C], catp[1][:, :, :C], sat=sat)
            ops.gemm_grouped_split(catp, lay.qkv_w, None, qkv, G, M, 3 * C, C, 2 * C, 3 * C, M * 2 * C, 3 * C * 2 * C, 0, M * 3 * C,
                                   w_scale=wsc(lay.qkv_w), sat=sat)
            ops.linattn_kv(qkv, kv, kvw, G, nimg, n, C)
            ops.linattn_apply(qkv, kv, msg, C, G, nimg, n, C)
            ops.split_planes(msg, msgp[0], msgp[1], sat=sat)
            ops.gemm_grouped_split(msgp, lay.merge_w, None, mrg, G, M, C, C, C, C, M * C, C * 2 * C, 0, M * C, w_scale=wsc(lay.merge_w), sat=sat)
            ops.layernorm(mrg, lay.n1w, lay.n1b, 1e-5, out=(catp[0][:, :, C:], catp[1][:, :, C:]), ldo=2 * C, rows_out=G * M,
                          rows_per_img=G * M, wgroup_rows=M, sat=sat)
            ops.gemm_grouped_split(catp, lay.mlp0_w, None, hidp, G, M, 2 * C, 2 * C, 2 * C, 2 * C, M * 2 * C, 2 * C * 2 * 2 * C, 0,
                                   M * 2 * C, act=ops.ACT_RELU, w_scale=wsc(lay.mlp0_w), sat=sat)
            ops.gemm_grouped_split(hidp, lay.mlp2_w, None, mrg, G, M, C, 2 * C, 2 * C, C, M * 2 * C, C * 2 * 2 * C, 0, M * C,
                                   w_scale=wsc(lay.mlp2_w), sat=sat)
        else:
            ops.gemm_grouped(cat, lay.qkv_w, None, qkv, G, M, 3 * C, C, 2 * C, C, 3 * C, M * 2 * C, 3 * C * C, 0, M * 3 * C)
            ops.linattn_kv(qkv, kv, kvw, G, nimg, n, C)
            ops.linattn_apply(qkv, kv, msg, C, G, nimg, n, C)
            fused_ln = C == 128 and lp != torch.float32   # 16-bit operands: Linear(-> 128) + LayerNorm in one pass (mk_gemm_ln128)
            if fused_ln:
                ops.gemm_ln128(msg, lay.merge_w, lay.n1w, lay.n1b, 1e-5, cat[:, :, C:], G, M, C, ldo=2 * C)
            else:
                ops.gemm_grouped(msg, lay.merge_w, None, mrg, G, M, C, C, C, C, C, M * C, C * C, 0, M * C)
                ops.layernorm(mrg, lay.n1w, lay.n1b, 1e-5, out=cat[:, :, C:], ldo=2 * C, rows_out=G * M, rows_per_img=G * M,
                              wgroup_rows=M)
            ops.gemm_grouped(cat, lay.mlp0_w, None, hid, G, M, 2 * C, 2 * C, 2 * C, 2 * C, 2 * C, M * 2 * C, 4 * C * C, 0,
                             M * 2 * C, act=ops.ACT_RELU)
            if not fused_ln:
                ops.gemm_grouped(hid, lay.mlp2_w, None, mrg, G, M, C, 2 * C, 2 * C, 2 * C, C, M * 2 * C, 2 * C * C, 0, M * C)
        if split:   # the layer's output as operand planes: of the next layer's `cat`, or (last) of resblock4's bordered input
            o2 = (x4h, x4l) if last else (catp[0][:, :, :C], catp[1][:, :, :C])
        else:
            o2 = x4 if last else cat
        if not split and C == 128 and lp != torch.float32:   # (fused_ln above): mlp[2] + norm2 + the layer's residual in one pass
            ops.gemm_ln128(hid, lay.mlp2_w, lay.n2w, lay.n2b, 1e-5, o2, G, M, 2 * C, ldo=C if last else 2 * C, resid=xs,
                           bordered=(nimg, gh, gw) if last else None)
        else:
            ops.layernorm(mrg, lay.n2w, lay.n2b, 1e-5, out=o2, ldo=C if last else 2 * C, resid=xs,
                          rows_out=G * M, rows_per_img=G * M, wgroup_rows=M, bordered=(nimg, gh, gw) if last else None,
                          sat=sat if split else None)
    # ---- resblock4 ----
    kpw, dw = W.rb4_kp, W.rb4_dsc
    ck = kpw.cout
    f4 = ws.get("rb4_f", (3, M, ck), torch.float32, dev)
    cd = dw.cout
    fd = ws.get("rb4_fd", (M, cd), torch.float32, dev)
    if split:
        h4p = plane_pair("rb4_h", (3, R, ck))
        ops.conv3x3_split((x4h[:3], x4l[:3]), C, kpw.w1, kpw.b1, h4p, ck, 3, nimg, gh, gw, act=ops.ACT_RELU, stride_in1=R * C, w_scale=wsc(kpw.w1), sat=sat,
                          stride_w=kpw.w1.shape[1] * kpw.w1.shape[2], stride_bias=ck, stride_out=R * ck, out_bordered=True)
        assert kpw.has_sc and dw.has_sc   # (weights.prepare gives the descriptor block identity shortcut columns in this mode)
        ops.conv3x3_split(h4p, ck, kpw.w2, kpw.b2, f4, ck, 3, nimg, gh, gw, act=ops.ACT_RELU, w_scale=wsc(kpw.w2),
                          in2=(x4h[:3], x4l[:3]), C2=C, stride_in1=R * ck, stride_in2=R * C,
                          stride_w=kpw.w2.shape[1] * kpw.w2.shape[2], stride_bias=ck, stride_out=M * ck)
        hdp = plane_pair("rb4_hd", (R, cd))
        ops.conv3x3_split((x4h[3], x4l[3]), C, dw.w1, dw.b1, hdp, cd, 1, nimg, gh, gw, act=ops.ACT_RELU, out_bordered=True, w_scale=wsc(dw.w1), sat=sat)
        ops.conv3x3_split(hdp, cd, dw.w2, dw.b2, fd, cd, 1, nimg, gh, gw, act=ops.ACT_NONE, w_scale=wsc(dw.w2),
                          in2=(x4h[3], x4l[3]), C2=C)   # relu=False, mickey_extractor.py:246
    else:
        h4 = ws.get("rb4_h" + geo, (3, R, ck), lp, dev, zero=True)
        hd = ws.get("rb4_hd" + geo, (R, cd), lp, dev, zero=True)
        ops.conv3x3(x4, C, kpw.w1, kpw.b1, h4, ck, 3, nimg, gh, gw, act=ops.ACT_RELU, stride_in1=R * C,
                    stride_w=kpw.w1.shape[1] * kpw.w1.shape[2], stride_bias=ck, stride_out=R * ck, out_bordered=True)
        ops.conv3x3(h4, ck, kpw.w2, kpw.b2, f4, ck, 3, nimg, gh, gw, act=ops.ACT_RELU,
                    in2=x4 if kpw.has_sc else None, C2=C, resid=None if kpw.has_sc else x4, stride_in1=R * ck,
                    stride_in2=R * C, stride_resid=R * C, stride_w=kpw.w2.shape[1] * kpw.w2.shape[2], stride_bias=ck,
                    stride_out=M * ck)
        ops.conv3x3(x4[3], C, dw.w1, dw.b1, hd, cd, 1, nimg, gh, gw, act=ops.ACT_RELU, out_bordered=True)
        ops.conv3x3(hd, cd, dw.w2, dw.b2, fd, cd, 1, nimg, gh, gw, act=ops.ACT_NONE,   # relu=False, mickey_extractor.py:246
                    in2=x4[3] if dw.has_sc else None, C2=C, resid=None if dw.has_sc else x4[3])
    kh = mk["KP_HEADS"]
    return ops.head_tails(f4[0], W.w_score, f4[1], W.w_xy, f4[2], W.w_depth, fd, nimg, gh, gw, ck, cd, border=3,
                          use_softmax=bool(kh["USE_SOFTMAX"]), use_depth_sigmoid=bool(kh["USE_DEPTHSIGMOID"]),
                          max_depth=float(kh["MAX_DEPTH"]), norm_dsc=bool(mk["DSC_HEAD"]["NORM_DSC"]),
                          down=float(mk["DINOV2"]["DOWN_FACTOR"]))


def match(W, cfg, dsc0, dsc1, scr0, scr1, lean=False):
    fm = cfg["FEATURE_MATCHER"]
    if fm["TYPE"] == "DualSoftmax":
        ds = fm["DUAL_SOFTMAX"]
        # AMD.MATCHER_CORR: 'auto' = the split-fp16 correlation on the 16-bit matrix cores when its preconditions hold
        # (L2-normalised 128-channel descriptors, mickey_hip.h: mk_dual_softmax_split), else the exact fp32 MFMA; 'fp32' | 'split16'
        mode = str(cfg["AMD"].get("MATCHER_CORR", "auto")).lower()
        can = bool(cfg["MICKEY"]["DSC_HEAD"]["NORM_DSC"]) and ops.dual_softmax_split_ok(dsc0.shape[1], float(ds["TEMPERATURE"]))
        if mode == "split16" and not can:
            raise ValueError("AMD.MATCHER_CORR: split16 needs MICKEY.DSC_HEAD.NORM_DSC, 128 descriptor channels and T >= 0.0145")
        return ops.dual_softmax(dsc0, dsc1, scr0, scr1, float(ds["TEMPERATURE"]), W.dustbin if ds["USE_DUSTBIN"] else None,
                                want_scores=not lean, want_kp=not lean, want_final=True, split=can and mode != "fp32")
    if fm["TYPE"] == "Sinkhorn":
        # the reference's Sinkhorn branch is unreachable through featureMatcher.forward; the maths
        # restated is feature_matcher.py:125-137 with matching_mat(dsc0, dsc1, None)
        alpha = W.dustbin if W.dustbin is not None else float(fm["SINKHORN"]["DUSTBIN_SCORE_INIT"])
        return ops.sinkhorn(dsc0, dsc1, alpha, int(fm["SINKHORN"]["NUM_IT"]), scr0, scr1, want_scores=not lean,
                            want_kp=not lean, want_final=True)
    raise ValueError("feature matcher not recognized: %r" % (fm["TYPE"],))


def solve(cfg, final_scores, kps0, depth0, kps1, depth1, K0, K1, seed=0, offset=0, noise_outer=None, noise_inner=None,
          idx3_in=None, debug=False, offset_dev=None, pair_base=0, ws=None):
    """reference probabilisticProcrustes.py:183-348 on device.  Returns a dict with R [B,3,3], t [B,1,3],
    inliers [B,1] and the intermediates needed for the inlier list.  pair_base = global index of pair 0 (keys the
    Philox streams: sharding a batch over calls / GPUs does not change any pair's draws)."""
    P = cfg["PROCRUSTES"]
    it_m, it_r, ns, k3 = int(P["IT_MATCHES"]), int(P["IT_RANSAC"]), int(P["NUM_SAMPLED_MATCHES"]), int(P["NUM_CORR_3D_3D"])
    if k3 != 3:
        raise ValueError("NUM_CORR_3D_3D must be 3 (the inference solver fits minimal 3-point samples)")
    B, n0, n1 = final_scores.shape
    dev = final_scores.device
    invalid = torch.zeros((1,), device=dev, dtype=torch.int32)
    # the sampler's workspace lives in the model's Workspace (ws) like every other buffer of a forward: zeroed once, self-cleaning
    # afterwards (mickey_hip.h); one model is one forward at a time
    work = None
    if ws is not None:
        key = ("exprace_work", B, it_m, ns, n0 * n1)
        work = ws.bufs.get(key)
        if work is None or work.device != dev:
            work = ws.bufs[key] = ops.exprace_work(B, it_m, ns, n0 * n1, dev)
    idx, cnt = ops.exprace_topk(final_scores.reshape(B, n0 * n1), it_m, ns, noise=noise_outer, seed=seed, offset=offset,
                                invalid=invalid, offset_dev=offset_dev, pair_base=pair_base, work=work)
    X, Y, w, corr = ops.gather_backproject(idx, final_scores, kps0, depth0, kps1, depth1, K0, K1, it_m)
    Rh, th, score, idx3 = ops.ransac_hypotheses(X, Y, w, it_r, float(P["TH_SOFT_INLIER"]), noise3=noise_inner,
                                                idx3_in=idx3_in, seed=seed, offset=offset + 1, offset_dev=offset_dev,
                                                set_base=pair_base * it_m)
    R, t, conf, best, mask, rounds, invalid = ops.refine_pose(X, Y, Rh, th, score, B, it_m, it_r, float(P["TH_INLIER"]),
                                                              int(P["NUM_REFINEMENTS"]), k3, invalid=invalid)
    out = {"R": R, "t": t, "inliers": conf, "best": best, "mask": mask, "corr": corr, "weights": w, "invalid": invalid,
           "it_ransac": it_r, "it_matches": it_m}
    if debug:
        out.update(idx=idx, cnt=cnt, X=X, Y=Y, R_hyp=Rh, t_hyp=th, score=score, idx3=idx3, rounds=rounds)
    return out


def inliers_list(sol):
    """Per pair [m_i, 7] = (u0, v0, u1, v1, score, d0, d1) of the final hard inliers, sorted by score
    (descending): reference probabilisticProcrustes.py:306-327.  Only built on request (demo 3-D
    visualisation); a handful of torch index ops on <= 2048 rows per pair, outside the hot path."""
    B = sol["R"].shape[0]
    if int(sol["invalid"].item()) != 0:
        return [torch.zeros([0, 5])] * B
    k = sol["mask"].shape[1]
    sets = sol["best"].long() // sol["it_ransac"] + torch.arange(B, device=sol["best"].device) * sol["it_matches"]
    out = []
    for b in range(B):
        sel = sol["mask"][b] != 0
        c = sol["corr"][sets[b]][sel]
        w = sol["weights"][sets[b]][sel]
        order = torch.argsort(w, descending=True)
        c, w = c[order], w[order]
        out.append(torch.cat([c[:, 0:4], w[:, None], c[:, 4:6]], 1))
    return out
