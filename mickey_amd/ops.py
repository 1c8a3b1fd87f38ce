"""Tensor-level wrappers over the C ABI (include/mickey_hip.h).  Each function allocates its outputs
with torch (device memory only), passes raw pointers + the current HIP stream to libmickey_hip.so and
returns tensors.  No arithmetic happens here."""

import torch


from ._native import ACT_GELU, ACT_NONE, ACT_RELU, call, dtype_code, ptr, query, stream  # noqa: F401

LOG2E = 1.4426950408889634


def _chk(t, dtype=None):
    assert t.is_cuda and t.is_contiguous(), "expected a contiguous device tensor"
    if dtype is not None:
        assert t.dtype == dtype, "expected %s, got %s" % (dtype, t.dtype)
    return t


def preprocess_u8(frames, H, W, out=None):
    """uint8 [n, Hs, Ws, 3] device frames -> fp32 [n, 3, H, W] in [0, 1] (resize + /255 + CHW, mk_preprocess_u8)."""
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3 and frames.is_contiguous()
    n, Hs, Ws, _ = frames.shape
    if out is None:
        out = torch.empty((n, 3, H, W), device=frames.device, dtype=torch.float32)
    call("mk_preprocess_u8", ptr(frames), Hs * Ws * 3, n, Hs, Ws, ptr(out), H, W, stream())
    return out


def gemm_set_tile(mode):
    """Dev knob (mickey_hip_dev.h): 0 auto, 1 force 128x128, 2 force 64x128, 7 force the 8-wave ping-pong, 10 force one wave
    per SIMD; 500 / 501: automatic use of the 64x128 tiling off / on."""
    call("mk_gemm_set_tile", int(mode))


def gemm(a, w, bias=None, act=ACT_NONE, out_f32=False, out=None, lda=None, K=None):
    """out = act(a[:, :K] @ w[:, :K].T + bias).  a [M, lda] lp, w [N, ldw] lp."""
    M = a.shape[0]
    lda = a.shape[1] if lda is None else lda
    N, ldw = w.shape
    K = min(lda, ldw) if K is None else K
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    call("mk_gemm", ptr(a), lda, ptr(w), ldw, ptr(bias), ptr(out), out.stride(0), M, N, K, act,
         int(out.dtype == torch.float32), dtype_code(a.dtype), stream())
    return out


def gemm_grouped(a, w, bias, out, groups, M, N, K, lda, ldw, ldc, stride_a, stride_w, stride_bias, stride_out, act=ACT_NONE):
    call("mk_gemm_grouped", ptr(a), lda, stride_a, ptr(w), ldw, stride_w, ptr(bias), stride_bias, ptr(out), ldc, stride_out,
         groups, M, N, K, act, int(out.dtype == torch.float32), dtype_code(a.dtype), stream())
    return out


def gemm_ls_residual(a, w, bias, gamma, x):
    """x += gamma * (a @ w.T + bias), x fp32 [M, N] in place."""
    M, K = a.shape
    N = w.shape[0]
    call("mk_gemm_ls_residual", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(gamma), ptr(x), x.stride(0), M, N, K,
         dtype_code(a.dtype), stream())
    return x


def gemm_qkv(a, w, bias, q, k, vt, nimg, ntok, ntok_pad, heads):
    qscale = (64.0 ** -0.5) * LOG2E
    call("mk_gemm_qkv", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(q), ptr(k), ptr(vt), nimg, ntok, ntok_pad,
         heads, qscale, dtype_code(a.dtype), stream())


# ---- LayerNorm folded into the GEMMs around it (mickey_hip.h: mk_gemm_*_ln) -------------------------------------------------
def gemm_ls_residual_ln(a, w, bias, gamma, xh, xl, stats, x_out=None, shift=None):
    """(xh + xl) += gamma * (a @ w.T + bias) on the split residual stream (two 16-bit planes, x = hi + lo); stats[m, N // 64, 2]
    receives the per-slot (sum, sum of squares) of the new fp32 rows.  With x_out (fp32 [M, N]) the new rows are written there
    instead and xh / xl / stats are left alone (last block).  shift (fp32 [M], as published by the consumer before): row
    centring, the rows become x + branch - shift (mickey_hip.h)."""
    M, K = a.shape
    N = w.shape[0]
    call("mk_gemm_ls_residual_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(gamma), ptr(xh), ptr(xl),
         xh.stride(0), ptr(stats), ptr(shift), ptr(x_out), x_out.stride(0) if x_out is not None else N, M, N, K,
         dtype_code(a.dtype), stream())


def gemm_patch_embed_ln(a, w, bias, pos, xh, xl, stats, nimg, npatch):
    D = w.shape[0]
    call("mk_gemm_patch_embed_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(pos), ptr(xh), ptr(xl), ptr(stats),
         nimg, npatch, D, w.shape[1], dtype_code(a.dtype), stream())


def cls_token_ln(cls, pos, xh, xl, stats, nimg, ntok, D):
    call("mk_cls_token_ln", ptr(cls), ptr(pos), ptr(xh), ptr(xl), ptr(stats), nimg, ntok, D, dtype_code(xh.dtype), stream())


def recentre_split(xh, xl, stats):
    """Every row of the split stream (xh + xl) minus its own mean, in place; stats rewritten (mk_recentre_split)."""
    rows, D = xh.shape
    call("mk_recentre_split", ptr(xh), ptr(xl), ptr(stats), rows, D, dtype_code(xh.dtype), stream())


def gemm_ln(a, w, bias, colsum, stats, eps, act=ACT_NONE, out=None, shift_out=None):
    """out = act(LN(x) @ W.T + b) with a = the hi plane of x, w = W * ln_weight, bias = b + W @ ln_bias (folded on the host).
    shift_out (fp32 [M]): receives every row's mean, for the next producer's row centring."""
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=a.dtype)
    call("mk_gemm_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(colsum), ptr(stats), float(eps), ptr(shift_out),
         ptr(out), out.stride(0), M, N, K, act, dtype_code(a.dtype), stream())
    return out


def gemm_qkv_ln(a, w, bias, colsum, stats, eps, q, k, vt, nimg, ntok, ntok_pad, heads, shift_out=None):
    qscale = (64.0 ** -0.5) * LOG2E
    call("mk_gemm_qkv_ln", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(colsum), ptr(stats), float(eps),
         ptr(shift_out), ptr(q), ptr(k), ptr(vt), nimg, ntok, ntok_pad, heads, qscale, dtype_code(a.dtype), stream())


def im2col_patch14(img, gh, gw, ldo, dtype, out=None):
    """img fp32 [nimg, 3, H, W] (any strides with unit innermost) -> [nimg*gh*gw, ldo] lp (out: rows to write into)."""
    assert img.dtype == torch.float32 and img.stride(3) == 1
    nimg = img.shape[0]
    if out is None:
        out = torch.empty((nimg * gh * gw, ldo), device=img.device, dtype=dtype)
    assert out.shape == (nimg * gh * gw, ldo) and out.dtype == dtype and out.is_contiguous()
    call("mk_im2col_patch14", ptr(img), img.stride(0), img.stride(1), img.stride(2), nimg, gh, gw, ptr(out), ldo,
         dtype_code(dtype), stream())
    return out


def gemm_patch_embed(a, w, bias, pos, x, nimg, npatch):
    D = w.shape[0]
    call("mk_gemm_patch_embed", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(pos), ptr(x), nimg, npatch, D,
         w.shape[1], dtype_code(a.dtype), stream())


def cls_token(cls, pos, x, nimg, ntok, D):
    call("mk_cls_token", ptr(cls), ptr(pos), ptr(x), nimg, ntok, D, stream())


def layernorm(x, w, b, eps, out=None, out_dtype=torch.bfloat16, resid=None, rows_out=None, rows_per_img=None, skip=0,
              ldo=None, wgroup_rows=0, ldx=None, ldr=None, bordered=None, sat=None):
    """LayerNorm rows of fp32 x [rows, D]; see mk_layernorm for the row remap and the residual form.
    bordered = (nimg, H, W): `out` is a stack of bordered feature maps (rows_out = k * nimg * H * W pixels).
    sat (plane outputs only): the saturation word (sat_word)."""
    D = w.shape[-1]
    rows_in = x.numel() // x.shape[-1]
    if rows_per_img is None:
        rows_per_img = rows_in
    if rows_out is None:
        rows_out = rows_in - skip * (rows_in // rows_per_img)
    if isinstance(out, (tuple, list)):   # (hi, lo) fp16 planes: the operand form of the split-operand head kernels
        oh, ol = out   # ol None: the hi plane alone = the rows rounded to fp16 (mk_layernorm_planes)
        assert oh.dtype == torch.float16 and (ol is None or (ol.dtype == torch.float16 and oh.stride() == ol.stride()))
        ldo = oh.stride(-2) if ldo is None else ldo
        bh, bw, bm = (bordered[1], bordered[2], bordered[0] * bordered[1] * bordered[2]) if bordered else (0, 0, 0)
        call("mk_layernorm_planes", ptr(x), x.stride(-2) if ldx is None else ldx, ptr(w), ptr(b), float(eps), ptr(oh), ptr(ol), ldo,
             SPLIT_ACT_SCALE, ptr(resid), (resid.stride(-2) if resid is not None else D) if ldr is None else ldr, rows_out, D,
             rows_per_img, skip, wgroup_rows, bh, bw, bm, sat_word(sat), stream())
        return out
    if out is None and out_dtype is not None:
        out = torch.empty((rows_out, D), device=x.device, dtype=out_dtype)
    is_f32 = out is not None and out.dtype == torch.float32
    ldo = (out.stride(-2) if out is not None else D) if ldo is None else ldo
    lp = out.dtype if (out is not None and not is_f32) else torch.bfloat16
    bh, bw, bm = (bordered[1], bordered[2], bordered[0] * bordered[1] * bordered[2]) if bordered else (0, 0, 0)
    call("mk_layernorm", ptr(x), x.stride(-2) if ldx is None else ldx, ptr(w), ptr(b), float(eps), ptr(out), ldo, int(is_f32),
         ptr(resid), (resid.stride(-2) if resid is not None else D) if ldr is None else ldr, rows_out, D, rows_per_img, skip,
         wgroup_rows, bh, bw, bm, dtype_code(lp), stream())
    return out


def gemm_ln128(a, w, ln_w, ln_b, eps, out, groups, M, K, lda=None, ldo=None, resid=None, bordered=None):
    """mk_gemm_ln128: out = LayerNorm(a[g] @ w[g]^T) * ln_w[g] + ln_b[g] (+ resid, updated in place) for 128 output features;
    a lp [groups, M, >= K], w lp [groups, 128, K], out lp (dense rows or, bordered = (nimg, H, W), bordered feature maps)."""
    assert a.dtype == w.dtype == out.dtype and w.shape[-2] == 128 and w.shape[-1] == K
    lda = a.stride(-2) if lda is None else lda
    ldo = out.stride(-2) if ldo is None else ldo
    bh, bw = (bordered[1], bordered[2]) if bordered else (0, 0)
    call("mk_gemm_ln128", ptr(a), lda, a.stride(0) if a.dim() == 3 else 0, ptr(w), w.stride(-2), w.stride(0) if w.dim() == 3 else 0, ptr(ln_w), ptr(ln_b),
         float(eps), ptr(resid), resid.stride(-2) if resid is not None else 128, ptr(out), ldo, groups, M, K, bh, bw, dtype_code(a.dtype),
         stream())
    return out


def attn_set_mode(mode):
    """Dev knob (mickey_hip_dev.h): attention kernel variant, 0 = automatic (tests / benchmarks)."""
    call("mk_attn_set_mode", int(mode))


def flash_attn(q, k, vt, out, nimg, heads, ntok, ntok_pad):
    call("mk_flash_attn_fwd", ptr(q), ptr(k), ptr(vt), ptr(out), out.stride(0), nimg, heads, ntok, ntok_pad,
         dtype_code(q.dtype), stream())
    return out


CONV_OUT_DENSE, CONV_OUT_BORDERED, CONV_OUT_F32 = 0, 1, 2


def bordered_rows(nimg, H, W):
    """Rows of a bordered feature map of nimg H x W grids (mickey_hip.h: mk_bordered_rows)."""
    return int(query("mk_bordered_rows", nimg, H, W))


def bordered_empty(lead, nimg, H, W, C, dtype, device):
    """Zeroed stack of bordered feature maps, [*lead, mk_bordered_rows, C]: the border rows stay zero for ever."""
    return torch.zeros(tuple(lead) + (bordered_rows(nimg, H, W), C), dtype=dtype, device=device)


def bordered_index(nimg, H, W, device):
    """Row of pixel (b, y, x) in a bordered feature map, int64 [nimg * H * W] (tests, tools)."""
    b, y, x = torch.meshgrid(torch.arange(nimg, device=device), torch.arange(H, device=device),
                             torch.arange(W, device=device), indexing="ij")
    return (((b * (H + 1) + y + 1) * (W + 1)) + x + 1).reshape(-1)


def conv3x3(in1, C1, w, bias, out, Cout, groups, nimg, H, W, act=ACT_NONE, in2=None, C2=0, resid=None,
            stride_in1=0, stride_in2=0, stride_w=0, stride_bias=0, stride_out=0, stride_resid=0, out_bordered=False):
    """in1 / in2 / resid: bordered feature maps; out: fp32 dense rows, or lp dense / bordered rows (mk_conv3x3)."""
    kind = CONV_OUT_F32 if out.dtype == torch.float32 and in1.dtype != torch.float32 else \
        (CONV_OUT_BORDERED if out_bordered else CONV_OUT_DENSE)
    call("mk_conv3x3", ptr(in1), stride_in1, C1, ptr(in2), stride_in2, C2, ptr(w), w.shape[-1], stride_w, ptr(bias), stride_bias,
         ptr(resid), stride_resid, ptr(out), Cout, stride_out, groups, nimg, H, W, act, kind, dtype_code(in1.dtype), stream())
    return out


SPLIT_ACT_SCALE, SPLIT_W_SCALE = 64.0, 1024.0   # powers of two: activations up to 1023, weights up to 63 stay finite in fp16


def sat_word(flag):
    """The per-call saturation word of the plane-writing entry points (mickey_hip.h: sat_flag): None, or a zero-initialised
    int32 [1] device tensor that a kernel ORs 1 into when it had to clamp |x * scale| at fp16's largest finite value or met a
    NaN.  Returns the device pointer (None = nobody watches: no per-element work)."""
    if flag is None:
        return None
    assert flag.dtype == torch.int32 and flag.numel() >= 1 and flag.is_cuda
    return ptr(flag)


def plane_pair(shape, device, zero=False):
    """(hi, lo) fp16 planes as the two halves of ONE allocation: the split-operand kernels address both planes of a source from
    one base pointer (mickey_hip.h: within 2 GiB of each other)."""
    t = (torch.zeros if zero else torch.empty)((2,) + tuple(shape), device=device, dtype=torch.float16)
    return t[0], t[1]


def split_planes(x, hi, lo, scale=SPLIT_ACT_SCALE, sat=None):
    """fp32 tensor -> fp16 planes with x * scale = hi + lo (mk_split_planes); hi / lo: preallocated, same shape.  x, hi, lo
    may be column blocks of wider row-major matrices (views whose last dimension is contiguous and whose rows are
    equidistant, e.g. t[..., :C]).  sat: the saturation word (sat_word)."""
    assert x.dtype == torch.float32 and hi.dtype == torch.float16 and lo.dtype == torch.float16 and hi.shape == x.shape == lo.shape
    cols = x.shape[-1]
    rows = x.numel() // cols

    def ld(t):   # row stride of a [..., cols] view with uniformly spaced rows
        assert t.stride(-1) == 1
        lead = t.stride(-2) if t.dim() > 1 else cols
        for d in range(t.dim() - 2):   # leading dims must continue the same row spacing
            assert t.stride(d) == t.stride(d + 1) * t.shape[d + 1], "rows must be equidistant"
        return lead
    assert ld(hi) == ld(lo)
    call("mk_split_planes", ptr(x), rows, cols, ld(x), float(scale), ptr(hi), ptr(lo), ld(hi), sat_word(sat), stream())
    return hi, lo


def gemm_grouped_split(a, w, bias, out, groups, M, N, K, lda, ldc, stride_a, stride_w, stride_bias, stride_out, act=ACT_NONE,
                       w_scale=SPLIT_W_SCALE, sat=None):
    """mk_gemm_grouped_split: a = (hi, lo) fp16 planes [groups, M, lda] (one allocation: plane_pair), w fp16 [groups, N, 2 K]
    (weights.split_conv_weight); out fp32 or a (hi, lo) pair."""
    ah, al = a
    assert w.dtype == torch.float16 and w.shape[-1] == 2 * K
    if isinstance(out, (tuple, list)):
        oh, ol = out
    else:
        oh, ol = out, None
        assert out.dtype == torch.float32
    call("mk_gemm_grouped_split", ptr(ah), ptr(al), lda, stride_a, ptr(w), 2 * K, stride_w, ptr(bias), stride_bias, ptr(oh), ptr(ol),
         ldc, stride_out, groups, M, N, K, act, 1.0 / (SPLIT_ACT_SCALE * float(w_scale)), SPLIT_ACT_SCALE, sat_word(sat), stream())
    return out


def conv3x3_split(in1, C1, w, bias, out, Cout, groups, nimg, H, W, act=ACT_NONE, in2=None, C2=0, stride_in1=0, stride_in2=0,
                  stride_w=0, stride_bias=0, stride_out=0, out_bordered=False, w_scale=SPLIT_W_SCALE, sat=None):
    """mk_conv3x3_split: in1 / in2 = (hi, lo) pairs of bordered fp16 planes (each pair one allocation: plane_pair), w fp16
    [.., Cout, 2 K] (weights.split_conv_weight); out: an fp32 tensor, or a (hi, lo) pair of fp16 planes = the operand form of
    the next split conv (no fp32 round trip, no mk_split_planes pass).  in1 = (hi, None): the source is its hi plane (fp16
    features of an fp16 encoder): two products instead of three (then without in2)."""
    assert w.dtype == torch.float16
    h1, l1 = in1
    h2, l2 = in2 if in2 is not None else (None, None)
    if isinstance(out, (tuple, list)):
        oh, ol = out
        assert oh.dtype == torch.float16 and ol.dtype == torch.float16 and oh.shape == ol.shape
    else:
        oh, ol = out, None
        assert out.dtype == torch.float32
    call("mk_conv3x3_split", ptr(h1), ptr(l1), stride_in1, C1, ptr(h2), ptr(l2), stride_in2, C2, ptr(w), w.shape[-1], stride_w,
         ptr(bias), stride_bias, ptr(oh), ptr(ol), Cout, stride_out, groups, nimg, H, W, act, int(out_bordered),
         1.0 / (SPLIT_ACT_SCALE * float(w_scale)), SPLIT_ACT_SCALE, sat_word(sat), stream())
    return out


def posenc_add(x, pe, xs, cat, groups, nimg, npix, C):
    call("mk_posenc_add", ptr(x), ptr(pe), ptr(xs), ptr(cat), cat.stride(-2), groups, nimg, npix, C, dtype_code(x.dtype),
         stream())


def linattn_work_floats(groups, nimg, L, C):
    return query("mk_linattn_work_floats", groups, nimg, L, C)


def linattn_kv(qkv, kv, work, groups, nimg, L, C):
    call("mk_linattn_kv", ptr(qkv), ptr(kv), ptr(work), groups, nimg, L, C, stream())


def linattn_apply(qkv, kv, out, ldo, groups, nimg, L, C):
    call("mk_linattn_apply", ptr(qkv), ptr(kv), ptr(out), ldo, groups, nimg, L, C, dtype_code(out.dtype), stream())


def head_tails(f_det, w_score, f_off, w_xy, f_dep, w_dep, f_dsc, nimg, h, w, C, Cd, border=3, use_softmax=True,
               use_depth_sigmoid=False, max_depth=60.0, norm_dsc=True, down=14.0):
    dev = f_det.device
    n = h * w
    scr = torch.empty((nimg, 1, n), device=dev, dtype=torch.float32)
    kps = torch.empty((nimg, 2, n), device=dev, dtype=torch.float32)
    depth = torch.empty((nimg, 1, n), device=dev, dtype=torch.float32)
    dsc = torch.empty((nimg, Cd, n), device=dev, dtype=torch.float32)
    call("mk_head_tails", ptr(f_det), ptr(w_score), ptr(f_off), ptr(w_xy), ptr(f_dep), ptr(w_dep), ptr(f_dsc), ptr(scr),
         ptr(kps), ptr(depth), ptr(dsc), nimg, h, w, C, Cd, border, int(use_softmax), int(use_depth_sigmoid), float(max_depth),
         int(norm_dsc), float(down), stream())
    return scr, kps, depth, dsc


def dual_softmax_split_ok(C, temperature):
    """Preconditions of mk_dual_softmax_split that can be checked on the host (the third, |dsc| <= 1, is the caller's)."""
    return C == 128 and temperature > 0 and LOG2E / float(temperature) <= 100.0


def dual_softmax(dsc0, dsc1, scr0=None, scr1=None, temperature=0.1, dustbin=None, want_scores=True, want_kp=True,
                 want_final=True, split=False):
    """Returns (scores, kp_scores, final_scores) (None where not requested).  split: the correlation on the 16-bit matrix
    cores with split-fp16 operands (mk_dual_softmax_split; unit-norm descriptors only), else the exact fp32 MFMA."""
    _chk(dsc0, torch.float32)
    _chk(dsc1, torch.float32)
    B, C, n0 = dsc0.shape
    n1 = dsc1.shape[2]
    dev = dsc0.device
    mk = lambda want: torch.empty((B, n0, n1), device=dev, dtype=torch.float32) if want else None  # noqa: E731
    scores = mk(want_scores)
    kp = mk(want_kp and scr0 is not None)
    fin = mk(want_final and scr0 is not None)
    if split:
        work = torch.empty((query("mk_dual_softmax_split_work_floats", B, n0, n1),), device=dev, dtype=torch.float32)
        call("mk_dual_softmax_split", ptr(dsc0), ptr(dsc1), ptr(scr0), ptr(scr1), 1.0 / float(temperature), int(dustbin is not None),
             float(dustbin) if dustbin is not None else 0.0, ptr(scores), ptr(kp), ptr(fin), ptr(work), B, C, n0, n1, stream())
        return scores, kp, fin
    work = torch.empty((query("mk_dual_softmax_work_floats", B, n0, n1, int(scores is None and fin is None)),), device=dev,
                       dtype=torch.float32)
    call("mk_dual_softmax", ptr(dsc0), ptr(dsc1), ptr(scr0), ptr(scr1), 1.0 / float(temperature), int(dustbin is not None),
         float(dustbin) if dustbin is not None else 0.0, ptr(scores), ptr(kp), ptr(fin), ptr(work), B, C, n0, n1, stream())
    return scores, kp, fin


def dual_softmax_set_chunks(chunks):
    """Dev knob (mickey_hip_dev.h): column chunks per row block in pass 2 of mk_dual_softmax_split (0 = default)."""
    call("mk_dual_softmax_set_chunks", int(chunks))


def sinkhorn_set_group(pairs):
    """Dev knob (mickey_hip_dev.h): pairs iterated together by mk_sinkhorn (0 = batch-wide, non-temporal reads: default)."""
    call("mk_sinkhorn_set_group", int(pairs))


def sinkhorn(dsc0, dsc1, alpha, iters=10, scr0=None, scr1=None, want_scores=True, want_kp=False, want_final=False):
    """Returns scores, or (scores, kp_scores, final_scores) when scr0/scr1 are given."""
    B, C, n0 = dsc0.shape
    n1 = dsc1.shape[2]
    dev = dsc0.device
    mk = lambda want: torch.empty((B, n0, n1), device=dev, dtype=torch.float32) if want else None  # noqa: E731
    out, kp, fin = mk(want_scores), mk(want_kp and scr0 is not None), mk(want_final and scr0 is not None)
    work = torch.empty((query("mk_sinkhorn_work_floats", B, n0, n1),), device=dev, dtype=torch.float32)
    call("mk_sinkhorn", ptr(dsc0), ptr(dsc1), ptr(scr0), ptr(scr1), float(alpha), int(iters), ptr(out), ptr(kp), ptr(fin),
         ptr(work), B, C, n0, n1, stream())
    return out if scr0 is None else (out, kp, fin)


def mutual_nn(scores):
    """Batched get_matches_list: returns (matches int32 [B, n0, 2], count int32 [B])."""
    _chk(scores, torch.float32)
    B, n0, n1 = scores.shape
    dev = scores.device
    matches = torch.zeros((B, n0, 2), device=dev, dtype=torch.int32)
    count = torch.zeros((B,), device=dev, dtype=torch.int32)
    work = torch.empty((2 * B * (n0 + n1),), device=dev, dtype=torch.int32)
    call("mk_mutual_nn", ptr(scores), ptr(matches), ptr(count), ptr(work), B, n0, n1, stream())
    return matches, count


# ---- solver ---------------------------------------------------------------------------------------

def counter_add(counter, inc):
    """counter (int64 CUDA tensor, 1 element) += inc on the current stream (see mk_counter_add)."""
    call("mk_counter_add", ptr(counter), int(inc), stream())


def exprace_set_mode(mode):
    """Dev knob (mickey_hip_dev.h): 0 = skip sampler (default), 1 = the pre-filter collect pass."""
    call("mk_exprace_set_mode", int(mode))


def dev_mfma_sustained(device, iters=400000, zero_operands=False, workgroups=None):
    """Measurement probe (mickey_hip_dev.h: mk_dev_mfma_sustained): TFLOP/s of back-to-back v_mfma_f32_16x16x32 (bf16) on
    register-resident operands, one 8-wave workgroup per CU, one warm launch + one timed launch (HIP events).  Not product code."""
    wg = workgroups or torch.cuda.get_device_properties(device).multi_processor_count
    scratch = torch.empty((wg * 512,), device=device, dtype=torch.float32)
    call("mk_dev_mfma_sustained", ptr(scratch), wg, max(1, iters // 8), int(zero_operands), stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call("mk_dev_mfma_sustained", ptr(scratch), wg, iters, int(zero_operands), stream())
    e1.record()
    e1.synchronize()
    return wg * iters * 1048576.0 / (e0.elapsed_time(e1) * 1e-3) / 1e12


def exprace_work(B, rows_per_pair, k, ncell, device):
    """Workspace of mk_exprace_topk with its self-cleaning state zeroed (mickey_hip.h): allocate ONCE per shape and stream and
    hand it to every exprace_topk call -- a call leaves the state zero, so no zero-fill launch stands in front of the chain."""
    work = torch.empty((query("mk_exprace_topk_work_bytes", B, rows_per_pair, k, ncell),), device=device, dtype=torch.uint8)
    work[:query("mk_exprace_topk_state_bytes", B, rows_per_pair)].zero_()
    return work


def exprace_topk(p, rows_per_pair, k, noise=None, seed=0, offset=0, invalid=None, offset_dev=None, pair_base=0, work=None):
    """p fp32 [B, ncell] -> (idx int32 [B*rows_per_pair, k], cnt int32 [B*rows_per_pair]).  work: exprace_work(...) kept by the
    caller across calls (None: allocated and zeroed here -- one more launch)."""
    p, noise = _c(p, noise)
    _chk(p, torch.float32)
    B, ncell = p.shape
    dev = p.device
    idx = torch.empty((B * rows_per_pair, k), device=dev, dtype=torch.int32)
    cnt = torch.empty((B * rows_per_pair,), device=dev, dtype=torch.int32)
    if work is None:
        work = exprace_work(B, rows_per_pair, k, ncell, dev)
    assert work.numel() >= query("mk_exprace_topk_work_bytes", B, rows_per_pair, k, ncell)
    call("mk_exprace_topk", ptr(p), ptr(noise), int(seed), int(offset), ptr(offset_dev), ptr(idx), ptr(cnt), ptr(invalid), ptr(work),
         B, rows_per_pair, ncell, k, int(pair_base), stream())
    return idx, cnt


def _c(*ts):
    """Device pointers are handed to C with implied dense row-major strides: normalise views."""
    return [None if t is None else t.contiguous() for t in ts]


def gather_backproject(idx, final_scores, kps0, depth0, kps1, depth1, K0, K1, rows_per_pair):
    idx, final_scores, kps0, depth0, kps1, depth1, K0, K1 = _c(idx, final_scores, kps0, depth0, kps1, depth1, K0, K1)
    B, n0, n1 = final_scores.shape
    R, k = idx.shape
    dev = idx.device
    X = torch.empty((R, k, 3), device=dev, dtype=torch.float32)
    Y = torch.empty((R, k, 3), device=dev, dtype=torch.float32)
    wts = torch.empty((R, k), device=dev, dtype=torch.float32)
    corr = torch.empty((R, k, 6), device=dev, dtype=torch.float32)
    call("mk_gather_backproject", ptr(idx), ptr(final_scores), ptr(kps0), ptr(depth0), ptr(kps1), ptr(depth1), ptr(K0),
         ptr(K1), ptr(X), ptr(Y), ptr(wts), ptr(corr), B, rows_per_pair, k, n0, n1, stream())
    return X, Y, wts, corr


def ransac_hypotheses(X, Y, wts, it_ransac, th_soft, noise3=None, idx3_in=None, seed=0, offset=0, offset_dev=None, set_base=0):
    X, Y, wts, noise3, idx3_in = _c(X, Y, wts, noise3, idx3_in)
    nsets, k, _ = X.shape
    dev = X.device
    nh = nsets * it_ransac
    Rh = torch.empty((nh, 9), device=dev, dtype=torch.float32)
    th = torch.empty((nh, 3), device=dev, dtype=torch.float32)
    score = torch.empty((nh,), device=dev, dtype=torch.float32)
    idx3 = torch.empty((nh, 3), device=dev, dtype=torch.int32)
    call("mk_ransac_hypotheses", ptr(X), ptr(Y), ptr(wts), ptr(noise3), ptr(idx3_in), int(seed), int(offset), ptr(offset_dev),
         float(th_soft), ptr(Rh), ptr(th), ptr(score), ptr(idx3), nsets, it_ransac, k, int(set_base), stream())
    return Rh, th, score, idx3


def train_ransac_masks(X, Y, wts, it_ransac, th_ref, num_ref, num_corr, noise=None, idx_in=None, seed=0, offset=0,
                       offset_dev=None, set_base=0):
    """Training-time hypotheses + refinement (reference loss_class.py:141-184): X, Y [nsets, S, 3], wts [nsets, S] ->
    (inliers_final fp32 [nsets*it_ransac, S], drawn indices int32 [nsets*it_ransac, num_corr], rounds int32)."""
    X, Y, wts, noise, idx_in = _c(X, Y, wts, noise, idx_in)
    _chk(X, torch.float32)
    nsets, S, _ = X.shape
    dev = X.device
    nh = nsets * it_ransac
    if idx_in is not None:
        idx_in = idx_in.to(torch.int32)
    mask = torch.empty((nh, S), device=dev, dtype=torch.float32)
    idx = torch.empty((nh, num_corr), device=dev, dtype=torch.int32)
    rounds = torch.empty((nh,), device=dev, dtype=torch.int32)
    call("mk_train_ransac_masks", ptr(X), ptr(Y), ptr(wts), ptr(noise), ptr(idx_in), int(seed), int(offset), ptr(offset_dev),
         float(th_ref), int(num_ref), int(num_corr), ptr(mask), ptr(idx), ptr(rounds), nsets, it_ransac, S, int(set_base),
         stream())
    return mask, idx, rounds


def reinforce_scatter(idx, loss_value, B, it_matches, ncell):
    """REINFORCE bookkeeping (reference loss_class.py:251-261): idx int32 [B*it_matches, S], loss_value fp32 [B*it_matches]
    -> (gradients, gradients_b) fp32 [B, ncell]."""
    idx, loss_value = _c(idx.to(torch.int32), loss_value.to(torch.float32))
    S = idx.shape[1]
    dev = idx.device
    grads = torch.zeros((B, ncell), device=dev, dtype=torch.float32)
    grads_b = torch.zeros((B, ncell), device=dev, dtype=torch.float32)
    call("mk_reinforce_scatter", ptr(idx), ptr(loss_value), ptr(grads), ptr(grads_b), B, it_matches, S, ncell, stream())
    return grads, grads_b

def train_tail_fwd(X, Y, mask, Rgt, tgt, K0, K1, it_ransac, it_matches, th_soft, loss_type, soft_clip, img_h=720.0):
    """Forward of the differentiable RANSAC tail (mk_train_tail_fwd): -> (out [nhyp, 4], Rt [nhyp, 12], saved [nhyp, 32])."""
    X, Y, mask, Rgt, tgt, K0, K1 = _c(X, Y, mask, Rgt, tgt, K0, K1)
    nsets, S, _ = X.shape
    nh = nsets * it_ransac
    dev = X.device
    out = torch.empty((nh, 4), device=dev, dtype=torch.float32)
    Rt = torch.empty((nh, 12), device=dev, dtype=torch.float32)
    saved = torch.zeros((nh, 32), device=dev, dtype=torch.float32)
    call("mk_train_tail_fwd", ptr(X), ptr(Y), ptr(mask), ptr(Rgt), ptr(tgt), ptr(K0), ptr(K1), nsets, it_ransac, S, it_matches,
         float(th_soft), int(loss_type), int(soft_clip), float(img_h), ptr(out), ptr(Rt), ptr(saved), stream())
    return out, Rt, saved


def train_tail_bwd(X, Y, mask, Rgt, tgt, K0, K1, it_ransac, it_matches, th_soft, loss_type, soft_clip, Rt, saved, grad_out,
                   img_h=720.0):
    """Backward of the same (mk_train_tail_bwd): grad_out [nhyp, 2] = dL/d(loss_value, score) -> (dL/dX, dL/dY) [nsets, S, 3]."""
    X, Y, mask, Rgt, tgt, K0, K1, grad_out = _c(X, Y, mask, Rgt, tgt, K0, K1, grad_out)
    nsets, S, _ = X.shape
    dev = X.device
    work = torch.empty((nsets * it_ransac, 16), device=dev, dtype=torch.float32)
    gX, gY = torch.empty_like(X), torch.empty_like(Y)
    call("mk_train_tail_bwd", ptr(X), ptr(Y), ptr(mask), ptr(Rgt), ptr(tgt), ptr(K0), ptr(K1), nsets, it_ransac, S, it_matches,
         float(th_soft), int(loss_type), int(soft_clip), float(img_h), ptr(Rt), ptr(saved), ptr(grad_out), ptr(work), ptr(gX),
         ptr(gY), stream())
    return gX, gY


def train_aggregate_fwd(out, Rt, saved, B, it_matches, it_ransac, temperature, add_null, null_loss, null_score):
    """mk_train_aggregate_fwd: -> (loss_value [B*it_matches], per_pair [B, 3] = sums of (loss_value, rot, trans) over a pair's sets,
    coef [nhyp, 2] for the backward, flags int32 [2] = (any non-finite R / t, number of rank-one hypotheses))."""
    out, Rt, saved = _c(out, Rt, saved)
    dev = out.device
    loss_value = torch.empty((B * it_matches,), device=dev, dtype=torch.float32)
    per_pair = torch.empty((B, 3), device=dev, dtype=torch.float32)
    coef = torch.empty((B * it_matches * it_ransac, 2), device=dev, dtype=torch.float32)
    flags = torch.zeros((2,), device=dev, dtype=torch.int32)
    call("mk_train_aggregate_fwd", ptr(out), ptr(Rt), ptr(saved), B, it_matches, it_ransac, float(temperature), int(bool(add_null)),
         float(null_loss), float(null_score), ptr(loss_value), ptr(per_pair), ptr(coef), ptr(flags), stream())
    return loss_value, per_pair, coef, flags


def train_aggregate_bwd(coef, g_pair, B, it_matches, it_ransac):
    """mk_train_aggregate_bwd: dL/d(pair baseline) [B] -> dL/d(loss_k, score_k) [nhyp, 2] (the grad_out of train_tail_bwd)."""
    coef, g_pair = _c(coef, g_pair.float())
    g = torch.empty_like(coef)
    call("mk_train_aggregate_bwd", ptr(coef), ptr(g_pair), B, it_matches, it_ransac, ptr(g), stream())
    return g


def gather_backproject_bwd(idx, corr, gX, gY, K0, K1, B, rows_per_pair, n0, n1):
    """mk_gather_backproject_bwd: dL/dX, dL/dY [B*rows_per_pair, k, 3] -> (dL/dkps0 [B,2,n0], dL/ddepth0 [B,1,n0], dL/dkps1, dL/ddepth1)."""
    idx, corr, gX, gY, K0, K1 = _c(idx, corr, gX, gY, K0, K1)
    k = idx.shape[1]
    buf = torch.zeros((B * 3 * (n0 + n1),), device=idx.device, dtype=torch.float32)   # one zero-fill for the four accumulators
    gk0, gd0 = buf[:2 * B * n0].view(B, 2, n0), buf[2 * B * n0:3 * B * n0].view(B, 1, n0)
    o = 3 * B * n0
    gk1, gd1 = buf[o:o + 2 * B * n1].view(B, 2, n1), buf[o + 2 * B * n1:].view(B, 1, n1)
    call("mk_gather_backproject_bwd", ptr(idx), ptr(corr), ptr(gX), ptr(gY), ptr(K0), ptr(K1), ptr(gk0), ptr(gd0), ptr(gk1), ptr(gd1),
         B, rows_per_pair, k, n0, n1, stream())
    return gk0, gd0, gk1, gd1


def refine_pose(X, Y, Rh, th, score, B, it_matches, it_ransac, th_inlier, num_ref, min_inliers, invalid=None):
    X, Y, Rh, th, score = _c(X, Y, Rh, th, score)
    k = X.shape[1]
    dev = X.device
    R = torch.empty((B, 3, 3), device=dev, dtype=torch.float32)
    t = torch.empty((B, 1, 3), device=dev, dtype=torch.float32)
    conf = torch.empty((B, 1), device=dev, dtype=torch.float32)
    best = torch.empty((B,), device=dev, dtype=torch.int32)
    mask = torch.empty((B, k), device=dev, dtype=torch.uint8)
    rounds = torch.empty((B,), device=dev, dtype=torch.int32)
    if invalid is None:
        invalid = torch.zeros((1,), device=dev, dtype=torch.int32)
    call("mk_refine_pose", ptr(X), ptr(Y), ptr(Rh), ptr(th), ptr(score), float(th_inlier), int(num_ref), int(min_inliers),
         ptr(R), ptr(t), ptr(conf), ptr(best), ptr(mask), ptr(rounds), ptr(invalid), B, it_matches, it_ransac, k, stream())
    call("mk_pose_finalize", ptr(R), ptr(t), ptr(conf), ptr(invalid), B, stream())
    return R, t, conf, best, mask, rounds, invalid
