"""-m gpu: each HIP kernel, called through the C ABI, against a plain fp32 torch reference of the same
op computed on the SAME (already 16-bit-rounded) inputs -- so the only differences are accumulation
order and the output rounding."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture(autouse=True)
def _auto_tile():
    yield
    if torch.cuda.is_available():
        from mickey_amd import ops
        ops.gemm_set_tile(0)
        ops.gemm_set_tile(400)   # tile order back to automatic
        ops.gemm_set_tile(602)   # persistent tile loop: the default (producers only)
        ops.attn_set_mode(0)


@pytest.mark.parametrize("M,N,K", [(3878, 3072, 1024), (5000, 1024, 512), (2100, 1280, 256)])
def test_gemm_tile_orders_are_bit_identical(M, N, K):
    """The 256x256 kernel's tile ORDER (bands of b m-tiles / groups of g n-tiles, automatic = n-fastest for outputs <= 4 tiles
    wide) only permutes which workgroup computes which tile: every order gives the same bits, including group widths that do not
    divide the number of n-tiles (12 = 8 + 4, 5 = 4 + 1, 5 = 3 + 2)."""
    from mickey_amd import ops
    dev = _dev()
    a = (torch.randn((M, K), generator=g(1)) * 0.5).bfloat16().to(dev)
    w = (torch.randn((N, K), generator=g(2)) / math.sqrt(K)).bfloat16().to(dev)
    bias = torch.randn((N,), generator=g(3)).to(dev)
    ops.gemm_set_tile(7)
    outs = {}
    for order in (400, 408, 403, 464 + 3, 464 + 4, 464 + 8, 464 + 12):
        ops.gemm_set_tile(order)
        outs[order] = ops.gemm(a, w, bias, act=ops.ACT_GELU, out_f32=False)
    ref = a.float() @ w.float().t() + bias
    assert rel(outs[400].float(), F.gelu(ref)) < 5e-3
    for order, o in outs.items():
        assert torch.equal(o, outs[400]), order


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("which", ["fc1", "qkv", "proj", "fc2", "last", "patch"])
def test_persistent_gemm_is_bit_identical(which, dtype):
    """The 256x256 kernel's PERSISTENT tile loop (round 5: one workgroup per CU walking a continuous K stream over its tiles, the
    next tile's first operands prefetched under the current epilogue, row parameters / shifts of the folded LayerNorm staged
    behind it; mk_gemm_set_tile 600 on / 601 off) against the one-tile-per-workgroup launch of the same kernel body: same tile
    order, same summation order, same epilogues -- every output bit for bit, on problems of 372 - 608 tiles (> 256 workgroups:
    each walks 1 - 3 tiles) with ragged last m-tiles, for the consumer (fc1 + GELU, qkv), producer (proj, fc2 with row centring;
    the last block's fp32 rows) and patch-embed forms of the encoder."""
    from mickey_amd import ops
    dev = _dev()
    gen = torch.Generator(device="cuda").manual_seed(11)
    rn = lambda *shape, s=1.0: torch.randn(shape, device=dev, generator=gen) * s  # noqa: E731
    D, heads, ntok, pad = 1024, 16, 1939, 1984
    ops.gemm_set_tile(7)

    def run(persist):
        ops.gemm_set_tile(600 if persist else 601)
        torch.manual_seed(0)
        if which in ("fc1", "qkv"):
            nimg = 4
            M, N = nimg * ntok, 4096 if which == "fc1" else 3072            # 31 m-tiles (last ragged) x 16 / 12
            x = RUN["x"]
            stats = RUN["stats"]
            shift = torch.full((M,), float("nan"), device=dev)
            if which == "fc1":
                out = ops.gemm_ln(x, RUN["w"], RUN["b"], RUN["colsum"], stats, 1e-6, act=ops.ACT_GELU, shift_out=shift)
                return [out, shift]
            q = torch.zeros((nimg, heads, pad, 64), device=dev, dtype=dtype)
            k = torch.zeros_like(q)
            vt = torch.zeros((nimg, heads, 64, pad), device=dev, dtype=dtype)
            ops.gemm_qkv_ln(x, RUN["w"], RUN["b"], RUN["colsum"], stats, 1e-6, q, k, vt, nimg, ntok, pad, heads, shift_out=shift)
            return [q, k, vt, shift]
        if which in ("proj", "fc2", "last"):
            M = 20 * ntok                                                   # 152 m-tiles (last ragged) x 4
            hi, lo = RUN["hi"].clone(), RUN["lo"].clone()
            st = torch.full((M, D // 64, 2), float("nan"), device=dev)
            x_out = torch.zeros((M, D), device=dev) if which == "last" else None
            ops.gemm_ls_residual_ln(RUN["a"], RUN["w"], RUN["b"], RUN["gamma"], hi, lo, st, x_out=x_out, shift=RUN["shift"])
            return [x_out] if which == "last" else [hi, lo, st]
        nimg, npatch = 20, 1938                                             # patch embed: K = 640 (10 stages), 152 x 4 tiles
        xh = torch.zeros((nimg * (npatch + 1), D), device=dev, dtype=dtype)
        xl = torch.zeros_like(xh)
        st = torch.zeros((nimg * (npatch + 1), D // 64, 2), device=dev)
        ops.gemm_patch_embed_ln(RUN["a"], RUN["w"], RUN["b"], RUN["pos"], xh, xl, st, nimg, npatch)
        return [xh, xl, st]

    RUN = {}
    if which in ("fc1", "qkv"):
        M, N = 4 * ntok, 4096 if which == "fc1" else 3072
        xf = rn(M, D) * (0.5 + 2 * torch.rand((M, 1), device=dev, generator=gen)) + rn(M, 1)
        RUN.update(x=xf.to(dtype), stats=_slot_stats(xf).float(), w=rn(N, D, s=1 / 32).to(dtype), b=rn(N, s=0.1), colsum=rn(N))
    elif which in ("proj", "fc2", "last"):
        M, K = 20 * ntok, 4096 if which == "fc2" else 1024
        xf = rn(M, D) * 2 + 0.7
        hi = xf.to(dtype)
        RUN.update(a=rn(M, K, s=0.5).to(dtype), w=rn(D, K, s=1 / math.sqrt(K)).to(dtype), b=rn(D), gamma=torch.rand((D,), device=dev, generator=gen),
                   hi=hi, lo=(xf - hi.float()).to(dtype), shift=rn(M, s=0.3))
    else:
        RUN.update(a=rn(20 * 1938, 640, s=0.5).to(dtype), w=rn(D, 640, s=0.04).to(dtype), b=rn(D, s=0.1), pos=rn(1939, D, s=0.1))
    one = run(False)
    per = run(True)
    again = run(True)
    for a_, b_, c_ in zip(one, per, again):
        assert bool(torch.isfinite(a_.float()).all())
        assert torch.equal(a_, b_) and torch.equal(b_, c_)
    ops.gemm_set_tile(602)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D,K,last", [(384, 384, False), (384, 1536, False), (384, 1536, True), (384, 256, False), (1024, 256, False)])
def test_persistent_gemm_is_bit_identical_short_k(D, K, last, dtype):
    """The persistent tile loop on the PRODUCER shapes the test above does not reach: ViT-S (D = 384: proj K = 384 = six K stages,
    fc2 K = 1536, the last block's fp32 rows; two n-tiles, the second ragged at 128 columns) and the shortest K stream the
    launcher sends there (K = 256 = four stages: the next tile's stage 0 is requested in the tile's third stage).  Persistent ==
    one tile per workgroup, bit for bit, run twice."""
    from mickey_amd import ops
    dev = _dev()
    gen = torch.Generator(device="cuda").manual_seed(13)
    rn = lambda *shape, s=1.0: torch.randn(shape, device=dev, generator=gen) * s  # noqa: E731
    M = 40 * 1939 if D == 384 else 20 * 1939       # 304 x 2 tiles / 152 x 4 tiles: more tiles than workgroups, ragged last m-tile
    xf = rn(M, D) * 2 + 0.7
    hi0 = xf.to(dtype)
    lo0 = (xf - hi0.float()).to(dtype)
    a, w = rn(M, K, s=0.5).to(dtype), rn(D, K, s=1 / math.sqrt(K)).to(dtype)
    b, gamma, shift = rn(D), torch.rand((D,), device=dev, generator=gen), rn(M, s=0.3)
    ops.gemm_set_tile(7)

    def run(persist):
        ops.gemm_set_tile(600 if persist else 601)
        hi, lo = hi0.clone(), lo0.clone()
        st = torch.full((M, D // 64, 2), float("nan"), device=dev)
        x_out = torch.zeros((M, D), device=dev) if last else None
        ops.gemm_ls_residual_ln(a, w, b, gamma, hi, lo, st, x_out=x_out, shift=shift)
        return [x_out] if last else [hi, lo, st]
    try:
        one, per, again = run(False), run(True), run(True)
        for a_, b_, c_ in zip(one, per, again):
            assert bool(torch.isfinite(a_.float()).all())
            assert torch.equal(a_, b_) and torch.equal(b_, c_)
    finally:
        ops.gemm_set_tile(602)
        ops.gemm_set_tile(0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K,last", [(1024, False), (4096, False), (4096, True), (192, False)])
def test_small_gemm_forms_are_bit_identical(K, last, dtype):
    """The producer GEMMs of ONE image pair (M = 3878: proj K = 1024, fc2 K = 4096, the last block's fp32 rows; K = 192 = three K
    tiles) on every tile form the launcher can pick: 128x128 with two LDS stages (mode 1), 64x128 with three (2), the 256x256
    ping-pong kernel (7) and the automatic choice (0).  All walk the K tiles in the same order through the same epilogue: every output bit for bit -- which is
    what keeps pair i of a batch equal to pair i alone whatever form its batch size selects."""
    from mickey_amd import ops
    dev = _dev()
    gen = torch.Generator(device="cuda").manual_seed(17)
    rn = lambda *shape, s=1.0: torch.randn(shape, device=dev, generator=gen) * s  # noqa: E731
    M, D = 2 * 1939, 1024
    xf = rn(M, D) * 2 + 0.7
    hi0 = xf.to(dtype)
    lo0 = (xf - hi0.float()).to(dtype)
    a, w = rn(M, K, s=0.5).to(dtype), rn(D, K, s=1 / math.sqrt(K)).to(dtype)
    b, gamma, shift = rn(D), torch.rand((D,), device=dev, generator=gen), rn(M, s=0.3)

    def run(tile):
        ops.gemm_set_tile(tile)
        hi, lo = hi0.clone(), lo0.clone()
        st = torch.full((M, D // 64, 2), float("nan"), device=dev)
        x_out = torch.zeros((M, D), device=dev) if last else None
        ops.gemm_ls_residual_ln(a, w, b, gamma, hi, lo, st, x_out=x_out, shift=shift)
        return [x_out] if last else [hi, lo, st]
    try:
        ref = run(1)
        for tile in (2, 7, 0, 2):
            for a_, b_ in zip(ref, run(tile)):
                assert bool(torch.isfinite(b_.float()).all())
                assert torch.equal(a_, b_), tile
    finally:
        ops.gemm_set_tile(0)


@pytest.mark.parametrize("tile", [1, 2, 7])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (3878, 3072, 1024), (129, 64, 576), (1000, 132, 64), (20000, 1024, 256)])
def test_gemm_bias_act(dtype, M, N, K, tile):
    from mickey_amd import ops
    dev = _dev()
    if dtype == torch.float32 and tile != 1:
        pytest.skip("the exact-fp32 mode has one schedule")
    ops.gemm_set_tile(tile)
    a = (torch.randn((M, K), generator=g(1)) * 0.5).to(dtype)
    w = (torch.randn((N, K), generator=g(2)) / math.sqrt(K)).to(dtype)
    bias = torch.randn((N,), generator=g(3))
    ref = a.float() @ w.float().t() + bias
    for act, fn in ((ops.ACT_NONE, lambda x: x), (ops.ACT_RELU, F.relu), (ops.ACT_GELU, F.gelu)):
        out32 = ops.gemm(a.to(dev), w.to(dev), bias.to(dev), act=act, out_f32=True)
        assert rel(out32, fn(ref)) < 2e-5, (act, rel(out32, fn(ref)))
        outlp = ops.gemm(a.to(dev), w.to(dev), bias.to(dev), act=act, out_f32=False)
        assert outlp.dtype == dtype
        assert rel(outlp.float(), fn(ref)) < {torch.bfloat16: 5e-3, torch.float16: 8e-4, torch.float32: 1e-5}[dtype]
    # asymmetric-operand transpose check: A = identity block picks rows of W
    eye = torch.zeros((64, K), dtype=dtype)
    eye[torch.arange(64), torch.arange(64)] = 1.0
    out = ops.gemm(eye.to(dev), w.to(dev), None, out_f32=True)
    assert torch.equal(out.cpu(), w.float()[:, :64].t().contiguous())


@pytest.mark.parametrize("tile", [0, 1, 2, 7])
@pytest.mark.parametrize("M,N,K", [(3878, 1024, 4096), (700, 384, 256)])
def test_gemm_ls_residual(M, N, K, tile):
    from mickey_amd import ops
    dev = _dev()
    ops.gemm_set_tile(tile)
    a = (torch.randn((M, K), generator=g(1)) * 0.5).bfloat16()
    w = (torch.randn((N, K), generator=g(2)) / math.sqrt(K)).bfloat16()
    bias, gamma = torch.randn((N,), generator=g(3)), torch.rand((N,), generator=g(4))
    x = torch.randn((M, N), generator=g(5))
    ref = x + gamma * (a.float() @ w.float().t() + bias)
    xd = x.to(dev)
    ops.gemm_ls_residual(a.to(dev), w.to(dev), bias.to(dev), gamma.to(dev), xd)
    assert rel(xd, ref) < 1e-5


@pytest.mark.parametrize("tile", [1, 2, 7])
def test_gemm_qkv_layout(tile):
    from mickey_amd import ops
    dev = _dev()
    ops.gemm_set_tile(tile)
    nimg, ntok, heads = 2, 333, 4
    D, pad = heads * 64, 384
    a = (torch.randn((nimg * ntok, D), generator=g(1)) * 0.5).bfloat16()
    w = (torch.randn((3 * D, D), generator=g(2)) / math.sqrt(D)).bfloat16()
    bias = torch.randn((3 * D,), generator=g(3))
    q = torch.zeros((nimg, heads, pad, 64), device=dev, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    vt = torch.zeros((nimg, heads, 64, pad), device=dev, dtype=torch.bfloat16)
    ops.gemm_qkv(a.to(dev), w.to(dev), bias.to(dev), q, k, vt, nimg, ntok, pad, heads)
    ref = (a.float() @ w.float().t() + bias).reshape(nimg, ntok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    qs = (64.0 ** -0.5) * ops.LOG2E
    assert rel(q[:, :, :ntok].float(), ref[0] * qs) < 5e-3
    assert rel(k[:, :, :ntok].float(), ref[1]) < 5e-3
    t = torch.arange(pad)
    perm = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    v_unperm = vt.cpu().float()[:, :, :, perm]  # position perm(t) holds token t
    assert rel(v_unperm[..., :ntok], ref[2].transpose(-1, -2)) < 5e-3
    assert float(q[:, :, ntok:].abs().sum()) == 0.0  # pad rows untouched


@pytest.mark.parametrize("tile", [0, 7])
@pytest.mark.parametrize("nimg,H,W,D", [(2, 75, 101, 256), (3, 300, 290, 384)])   # 5 x 7 and 21 x 20 patches
def test_patch_embed_and_cls(nimg, H, W, D, tile):
    from mickey_amd import ops
    dev = _dev()
    ops.gemm_set_tile(tile)
    gh, gw = H // 14, W // 14
    img = torch.rand((nimg, 3, H, W), generator=g(1))
    wconv = torch.randn((D, 3, 14, 14), generator=g(2)) / math.sqrt(588)
    bias = torch.randn((D,), generator=g(3)) * 0.1
    pos = torch.randn((1 + gh * gw, D), generator=g(4)) * 0.1
    cls = torch.randn((D,), generator=g(5)) * 0.1
    a = ops.im2col_patch14(img.to(dev), gh, gw, 640, torch.bfloat16)
    w2 = torch.zeros((D, 640))
    w2[:, :588] = wconv.reshape(D, 588)
    x = torch.zeros((nimg, 1 + gh * gw, D), device=dev)
    ops.gemm_patch_embed(a, w2.bfloat16().to(dev), bias.to(dev), pos.to(dev), x, nimg, gh * gw)
    ops.cls_token(cls.to(dev), pos.to(dev), x, nimg, 1 + gh * gw, D)
    crop = img[:, :, : gh * 14, : gw * 14].bfloat16().float()
    ref = F.conv2d(crop, wconv.bfloat16().float(), bias, stride=14).flatten(2).transpose(1, 2) + pos[1:]
    assert rel(x[:, 1:], ref) < 1e-5
    assert rel(x[:, 0], (cls + pos[0]).expand(nimg, D)) < 1e-7


def _slot_stats(x):
    """(sum, sum of squares) of every row over 64-column slots: what the producer epilogues emit."""
    M, N = x.shape
    xs = x.double().reshape(M, N // 64, 64)
    return torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1)


def _split(x, dtype):
    """fp32 -> the two 16-bit planes of the split residual stream."""
    hi = x.to(dtype)
    return hi, (x - hi.float()).to(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [1, 2, 7])
@pytest.mark.parametrize("M,N,K", [(3878, 1024, 256), (700, 384, 128), (257, 128, 64)])
def test_ln_fold_producer_residual(M, N, K, tile, dtype):
    """mk_gemm_ls_residual_ln on the split residual stream (x = hi + lo): the new fp32 rows are what mk_gemm_ls_residual
    computes from hi + lo (bit for bit, checked through the fp32 output form), hi / lo are their two-step rounding, the
    per-slot statistics are those of the fp32 rows."""
    from mickey_amd import ops
    dev = _dev()
    a = (torch.randn((M, K), generator=g(1)) * 0.5).to(dtype).to(dev)
    w = (torch.randn((N, K), generator=g(2)) / math.sqrt(K)).to(dtype).to(dev)
    bias, gamma = torch.randn((N,), generator=g(3)).to(dev), torch.rand((N,), generator=g(4)).to(dev)
    hi0, lo0 = _split((torch.randn((M, N), generator=g(5)) * 2 + 0.7).to(dev), dtype)
    ops.gemm_set_tile(tile)
    x_ref = ops.gemm_ls_residual(a, w, bias, gamma, hi0.float() + lo0.float())
    hi, lo = hi0.clone(), lo0.clone()
    stats = torch.full((M, N // 64, 2), float("nan"), device=dev)
    ops.gemm_ls_residual_ln(a, w, bias, gamma, hi, lo, stats)
    hr, lr = _split(x_ref, dtype)
    assert torch.equal(hi, hr) and torch.equal(lo, lr)
    # the pair carries 2 x the mantissa of one plane
    assert rel(hi.float() + lo.float(), x_ref) < (2e-5 if dtype == torch.bfloat16 else 5e-7)
    ref = _slot_stats(x_ref)
    assert rel(stats[..., 0], ref[..., 0]) < 1e-5 and rel(stats[..., 1], ref[..., 1]) < 1e-5
    assert bool(torch.isfinite(stats).all())
    # one summation order in every epilogue (quad_stats in mk_gemm_common.hpp): the statistics of a row do not depend on the
    # schedule, the tile shape or where in a tile the row sits -- which is what keeps pair i of a 32-pair batch bit-identical
    # to the same pair run alone
    ops.gemm_set_tile(1)
    hi1, lo1, st1 = hi0.clone(), lo0.clone(), torch.zeros_like(stats)
    ops.gemm_ls_residual_ln(a, w, bias, gamma, hi1, lo1, st1)
    assert torch.equal(st1, stats) and torch.equal(hi1, hi) and torch.equal(lo1, lo)
    if M > 300:   # the same rows at another tile alignment (first 37 rows dropped)
        hi3, lo3, st3 = hi0[37:].clone(), lo0[37:].clone(), torch.zeros_like(stats[37:])
        ops.gemm_set_tile(tile)
        ops.gemm_ls_residual_ln(a[37:].contiguous(), w, bias, gamma, hi3, lo3, st3)
        assert torch.equal(st3, stats[37:]) and torch.equal(hi3, hi[37:])
    ops.gemm_set_tile(tile)
    # last-block form: fp32 rows out, planes and statistics untouched
    hi2, lo2, st2 = hi0.clone(), lo0.clone(), torch.zeros_like(stats)
    x_out = torch.zeros((M, N), device=dev)
    ops.gemm_ls_residual_ln(a, w, bias, gamma, hi2, lo2, st2, x_out=x_out)
    assert torch.equal(x_out, x_ref) and torch.equal(hi2, hi0) and torch.equal(lo2, lo0) and float(st2.abs().sum()) == 0.0


@pytest.mark.parametrize("tile", [0, 7])
@pytest.mark.parametrize("nimg,H,W,D", [(2, 75, 101, 256), (3, 300, 290, 384)])
def test_ln_fold_producer_patch_embed_and_cls(nimg, H, W, D, tile):
    from mickey_amd import ops
    dev = _dev()
    ops.gemm_set_tile(tile)
    gh, gw = H // 14, W // 14
    ntok = 1 + gh * gw
    img = torch.rand((nimg, 3, H, W), generator=g(1))
    w2 = torch.zeros((D, 640))
    w2[:, :588] = torch.randn((D, 588), generator=g(2)) / math.sqrt(588)
    w2 = w2.bfloat16().to(dev)
    bias, pos = (torch.randn((D,), generator=g(3)) * 0.1).to(dev), (torch.randn((ntok, D), generator=g(4)) * 0.1).to(dev)
    cls = (torch.randn((D,), generator=g(5)) * 0.1).to(dev)
    a = ops.im2col_patch14(img.to(dev), gh, gw, 640, torch.bfloat16)
    x_ref = torch.zeros((nimg, ntok, D), device=dev)
    ops.gemm_patch_embed(a, w2, bias, pos, x_ref, nimg, gh * gw)
    ops.cls_token(cls, pos, x_ref, nimg, ntok, D)
    xh = torch.zeros((nimg * ntok, D), device=dev, dtype=torch.bfloat16)
    xl = torch.zeros_like(xh)
    stats = torch.full((nimg * ntok, D // 64, 2), float("nan"), device=dev)
    ops.gemm_patch_embed_ln(a, w2, bias, pos, xh, xl, stats, nimg, gh * gw)
    ops.cls_token_ln(cls, pos, xh, xl, stats, nimg, ntok, D)
    hr, lr = _split(x_ref.reshape(-1, D), torch.bfloat16)
    assert torch.equal(xh, hr) and torch.equal(xl, lr)
    ref = _slot_stats(x_ref.reshape(-1, D))
    assert rel(stats[..., 0], ref[..., 0]) < 1e-5 and rel(stats[..., 1], ref[..., 1]) < 1e-5


def _ln_fold_inputs(M, D, N, dtype, dev):
    """Rows with a mean well away from zero, a spread of scales and a few outlier channels (what DINOv2's residual stream
    looks like), LayerNorm affine parameters, a linear layer; returns what the host-side fold produces."""
    x = torch.randn((M, D), generator=g(1)) * (0.5 + 3 * torch.rand((M, 1), generator=g(2))) + 1.5 * torch.randn((M, 1), generator=g(3))
    x[:, 7] += 40.0
    x[:, D - 3] -= 25.0
    lw, lb = 1 + 0.3 * torch.randn((D,), generator=g(4)), 0.2 * torch.randn((D,), generator=g(5))
    W, b = torch.randn((N, D), generator=g(6)) / math.sqrt(D), torch.randn((N,), generator=g(7)) * 0.1
    wf = (W * lw).to(dtype)                       # W . diag(ln_weight), rounded once
    colsum = wf.float().sum(1)
    bf = b + W @ lb
    xlp = x.to(dtype)
    stats = _slot_stats(x).float()
    mean = x.double().mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(x.double().var(1, unbiased=False, keepdim=True) + 1e-6)
    # the arithmetic the kernel performs, in fp64 on the same rounded operands
    y_same = rstd * (xlp.double() @ wf.double().t()) - rstd * mean * colsum.double() + bf.double()
    # what the reference computes: LayerNorm in fp32, then the linear layer
    y_true = F.layer_norm(x.double(), (D,), lw.double(), lb.double(), 1e-6) @ W.double().t() + b.double()
    t = lambda v: v.to(dev)  # noqa: E731
    return t(xlp), t(wf), t(bf), t(colsum), t(stats), y_same, y_true


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [1, 2, 7])
@pytest.mark.parametrize("M,D,N,act", [(3878, 1024, 1024, 2), (700, 384, 512, 0), (130, 128, 256, 2)])
def test_ln_fold_consumer_gemm(M, D, N, act, tile, dtype):
    """mk_gemm_ln == act(LayerNorm(x) @ W^T + b) with the normalisation applied in the epilogue: exact (fp32 accumulation)
    against the same formula on the same rounded operands, and at the operand type's noise floor against the fp64 LN."""
    from mickey_amd import ops
    dev = _dev()
    xlp, wf, bf, colsum, stats, y_same, y_true = _ln_fold_inputs(M, D, N, dtype, dev)
    ops.gemm_set_tile(tile)
    out = ops.gemm_ln(xlp, wf, bf, colsum, stats, 1e-6, act=act)
    f = (lambda v: F.gelu(v)) if act == 2 else (lambda v: v)
    lp_eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert rel(out.float(), f(y_same)) < 0.6 * lp_eps            # output rounding only
    e = rel(out.float(), f(y_true))
    print("ln-fold consumer", dtype, (M, D, N), "vs fp64 LayerNorm + linear: %.2e" % e)
    assert e < 3 * lp_eps


def _dino_like_rows(M, D, level, gen):
    """Rows of a DINOv2-like residual stream: unit-scale channels with a spread of row scales, a common-mode level of
    `level` standard deviations per token, 'massive activation' channels (+-300..600 in 3 fixed channels on 2 % of the
    tokens), and LayerNorm weights with a few large entries."""
    sig = 0.5 + 2.0 * torch.rand((M, 1), generator=gen)
    x = torch.randn((M, D), generator=gen) * sig + level * sig * (1 + 0.2 * torch.randn((M, 1), generator=gen))
    hot = torch.rand((M,), generator=gen) < 0.02
    for ch, amp in ((5, 450.0), (D // 2 + 3, -600.0), (D - 7, 300.0)):
        x[hot, ch] += amp * (0.8 + 0.4 * torch.rand((int(hot.sum()),), generator=gen))
    lw = 1 + 0.3 * torch.randn((D,), generator=gen)
    lw[torch.tensor([5, 17, D - 7])] = torch.tensor([6.0, -4.0, 5.0])
    lb = 0.2 * torch.randn((D,), generator=gen)
    return x, lw, lb


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [1, 7])
@pytest.mark.parametrize("level", [0.0, 3.0, 30.0])
def test_ln_fold_row_centring_chain(level, tile, dtype):
    """The folded LayerNorm on rows with a common-mode level of 0 / 3 / 30 sigma and massive-activation channels
    (block.py:84-88: x + ls(f(norm(x))); what the residual stream of released DINOv2 weights looks like is not knowable
    offline, so the mechanism is tested at levels far beyond anything plausible).  Chain: consumer 1 on the raw rows (it
    publishes the row means) -> producer with shift_in (rows become x + branch - mean) -> consumer 2.  Consumer 2 must sit at
    the operand type's noise floor (3 x 2^-8 bf16 / 3 x 2^-11 fp16) against fp64 LayerNorm + linear of x + branch AT EVERY
    LEVEL; consumer 1 (un-centred rows: the only GEMM of a forward that can meet them is the first qkv, on the patch
    embedding's output) is reported, and asserted at the levels where raw rows are still fine."""
    from mickey_amd import ops
    dev = _dev()
    M, D, N, K = 1500, 1024, 512, 128
    gen = g(int(level) + 11)
    x0, lw, lb = _dino_like_rows(M, D, level, gen)
    W, b = torch.randn((N, D), generator=gen) / math.sqrt(D), torch.randn((N,), generator=gen) * 0.1
    wf = (W * lw).to(dtype)
    colsum, bf = wf.float().sum(1), b + W @ lb
    hi0, lo0 = _split(x0.to(dev), dtype)
    st0 = _slot_stats((hi0.float() + lo0.float()).cpu()).float().to(dev)
    ops.gemm_set_tile(tile)
    lp_eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11

    def ln_lin(x):
        return F.layer_norm(x.double(), (D,), lw.double(), lb.double(), 1e-6) @ W.double().t() + b.double()
    shift = torch.full((M,), float("nan"), device=dev)
    y1 = ops.gemm_ln(hi0, wf.to(dev), bf.to(dev), colsum.to(dev), st0, 1e-6, shift_out=shift)
    x0s = (hi0.float() + lo0.float()).cpu()
    assert rel(shift, x0s.double().mean(1)) < 1e-5
    e1 = rel(y1.float(), ln_lin(x0s))
    # producer: x1 = x0 + gamma * (a @ w^T + bias) - shift
    a = (torch.randn((M, K), generator=gen) * 0.5).to(dtype).to(dev)
    w2 = (torch.randn((D, K), generator=gen) / math.sqrt(K)).to(dtype).to(dev)
    bias2, gamma = torch.randn((D,), generator=gen).to(dev), torch.rand((D,), generator=gen).to(dev)
    hi, lo = hi0.clone(), lo0.clone()
    st1 = torch.full((M, D // 64, 2), float("nan"), device=dev)
    ops.gemm_ls_residual_ln(a, w2, bias2, gamma, hi, lo, st1, shift=shift)
    branch = gamma.cpu().double() * (a.cpu().double() @ w2.cpu().double().t() + bias2.cpu().double())
    x1 = x0s.double() + branch                     # what the reference's stream holds
    x1c = x1 - shift.cpu().double()[:, None]       # what the centred stream holds
    got = hi.float().cpu().double() + lo.float().cpu().double()
    assert rel(got, x1c) < (3e-5 if dtype == torch.bfloat16 else 1e-6)
    assert float((got.mean(1).abs() / got.std(1)).max()) < 1.5     # centred to within one update
    ref_st = _slot_stats(got.float())
    assert rel(st1[..., 0], ref_st[..., 0]) < 1e-3 and rel(st1[..., 1], ref_st[..., 1]) < 1e-5
    y2 = ops.gemm_ln(hi, wf.to(dev), bf.to(dev), colsum.to(dev), st1, 1e-6, shift_out=shift)
    e2 = rel(y2.float(), ln_lin(x1))               # LayerNorm of the UN-shifted rows: the shift is invisible
    print("ln-fold centring level %.0f %s tile %d: un-centred rows %.2e, centred rows %.2e (floor %.2e)" % (level, dtype, tile, e1, e2, 3 * lp_eps))
    assert e2 < 3 * lp_eps, (e1, e2)
    if level <= 3.0:
        assert e1 < 3 * lp_eps * (1 + level), e1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [1024, 384])
def test_recentre_split(D, dtype):
    """mk_recentre_split: every row minus its own mean, as two 16-bit planes, statistics of the centred fp32 rows."""
    from mickey_amd import ops
    dev = _dev()
    M = 777
    x, _, _ = _dino_like_rows(M, D, 30.0, g(5))
    hi, lo = _split(x.to(dev), dtype)
    x0 = hi.float().cpu().double() + lo.float().cpu().double()
    st = torch.full((M, D // 64, 2), float("nan"), device=dev)
    ops.recentre_split(hi, lo, st)
    got = hi.float().cpu().double() + lo.float().cpu().double()
    want = x0 - x0.mean(1, keepdim=True)
    assert rel(got, want) < (3e-5 if dtype == torch.bfloat16 else 1e-6)
    assert float(got.mean(1).abs().max()) < 1e-2 * float(got.std(1).min())
    ref = _slot_stats(got.float())
    assert rel(st[..., 1], ref[..., 1]) < 1e-5 and float((st[..., 0].cpu().double() - ref[..., 0]).abs().max()) < 1e-2
    assert bool(torch.isfinite(st).all())


@pytest.mark.parametrize("tile", [1, 7])
def test_ln_fold_consumer_qkv(tile):
    from mickey_amd import ops
    dev = _dev()
    nimg, ntok, heads = 2, 333, 4
    D, pad = heads * 64, 384
    xlp, wf, bf, colsum, stats, y_same, _ = _ln_fold_inputs(nimg * ntok, D, 3 * D, torch.bfloat16, dev)
    q = torch.zeros((nimg, heads, pad, 64), device=dev, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    vt = torch.zeros((nimg, heads, 64, pad), device=dev, dtype=torch.bfloat16)
    ops.gemm_set_tile(tile)
    ops.gemm_qkv_ln(xlp, wf, bf, colsum, stats, 1e-6, q, k, vt, nimg, ntok, pad, heads)
    ref = y_same.reshape(nimg, ntok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    qs = (64.0 ** -0.5) * ops.LOG2E
    assert rel(q[:, :, :ntok].float(), ref[0] * qs) < 3e-3
    assert rel(k[:, :, :ntok].float(), ref[1]) < 3e-3
    t = torch.arange(pad)
    perm = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    assert rel(vt.cpu().float()[:, :, :, perm][..., :ntok], ref[2].transpose(-1, -2)) < 3e-3
    assert float(q[:, :, ntok:].abs().sum()) == 0.0


@pytest.mark.parametrize("D", [128, 384, 1024])
def test_layernorm(D):
    from mickey_amd import ops
    dev = _dev()
    nimg, ntok = 3, 50
    x = torch.randn((nimg * ntok, D), generator=g(1)) * 3 + 0.5
    w, b = torch.randn((D,), generator=g(2)), torch.randn((D,), generator=g(3))
    ref = F.layer_norm(x, (D,), w, b, 1e-6)
    out = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6, out_dtype=torch.float32)
    assert rel(out, ref) < 2e-6
    out16 = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6, out_dtype=torch.bfloat16)
    assert rel(out16.float(), ref) < 4e-3
    # drop the first row of every image (CLS)
    outs = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6, out_dtype=torch.float32, rows_per_img=ntok, skip=1)
    assert rel(outs, ref.reshape(nimg, ntok, D)[:, 1:].reshape(-1, D)) < 2e-6
    # residual form
    r = torch.randn((nimg * ntok, D), generator=g(4))
    rd = r.to(dev)
    o2 = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-5, out_dtype=torch.bfloat16, resid=rd)
    ref2 = r + F.layer_norm(x, (D,), w, b, 1e-5)
    assert rel(rd, ref2) < 2e-6 and rel(o2.float(), ref2) < 4e-3


@pytest.mark.parametrize("D", [64, 100, 128, 384])
def test_layernorm_groups_and_bordered_output(D):
    """the forms the heads use: one (w, b) per block of rows, residual update, output into a stack of bordered feature maps
    (border rows untouched); D <= 128 takes the two-rows-per-wave kernel, ragged row counts included"""
    from mickey_amd import ops
    dev = _dev()
    G, nimg, H, W = 3, 2, 5, 7
    M = nimg * H * W
    x = torch.randn((G * M, D), generator=g(1)) * 2 - 0.3
    w, b = torch.randn((G, D), generator=g(2)), torch.randn((G, D), generator=g(3))
    r = torch.randn((G * M, D), generator=g(4))
    ref = torch.cat([F.layer_norm(x[i * M:(i + 1) * M], (D,), w[i], b[i], 1e-5) for i in range(G)]) + r
    R = ops.bordered_rows(nimg, H, W)
    for dt, tol in ((torch.float32, 2e-6), (torch.bfloat16, 4e-3)):
        out = torch.full((G, R, D), 7.0, device=dev, dtype=dt)
        rd = r.to(dev)
        ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-5, out=out, ldo=D, resid=rd, rows_out=G * M, rows_per_img=G * M,
                      wgroup_rows=M, bordered=(nimg, H, W))
        idx = ops.bordered_index(nimg, H, W, dev)
        mask = torch.ones(R, dtype=torch.bool, device=dev)
        mask[idx] = False
        assert bool((out[:, mask] == 7.0).all())
        assert rel(out[:, idx].reshape(G * M, D).float(), ref) < tol and rel(rd, ref) < 2e-6
    # ragged: 13 rows (not a multiple of the rows a wave owns)
    o = ops.layernorm(x[:13].to(dev), w[0].to(dev), b[0].to(dev), 1e-6, out_dtype=torch.float32)
    assert rel(o, F.layer_norm(x[:13], (D,), w[0], b[0], 1e-6)) < 2e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("ntok,nimg,heads", [(64, 2, 3), (200, 2, 3), (1939, 2, 3), (1939, 8, 8), (300, 32, 16)])
def test_flash_attention(dtype, ntok, nimg, heads, mode):
    """(1939, 8, 8) and (300, 32, 16) are large enough grids to take the 64-queries-per-wave instantiation in auto mode;
    modes 1 / 2 = the production kernel (lean softmax) with 32 / 64 queries per wave, 3 = the classic online softmax,
    4 / 5 = the production kernel's structure on the 16x16x32 MFMA."""
    from mickey_amd import ops
    dev = _dev()
    ops.attn_set_mode(mode)
    D = heads * 64
    pad = (ntok + 63) // 64 * 64
    qkv = torch.randn((3, nimg, heads, ntok, 64), generator=g(ntok)) * 1.5
    qkv[0, 0, 0, 5] *= 6.0   # a spiky query row: forces large running-max jumps
    qkv[1, 0, 0, 130 % ntok] *= 6.0
    q16, k16, v16 = (qkv[0] * 0.125 * ops.LOG2E).to(dtype), qkv[1].to(dtype), qkv[2].to(dtype)
    q = torch.zeros((nimg, heads, pad, 64), dtype=dtype)
    k = torch.zeros_like(q)
    vt = torch.zeros((nimg, heads, 64, pad), dtype=dtype)
    q[:, :, :ntok], k[:, :, :ntok] = q16, k16
    t = torch.arange(ntok)
    perm = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    vt[:, :, :, perm] = v16.transpose(-1, -2)
    out = torch.zeros((nimg * ntok, D), device=dev, dtype=dtype)
    ops.flash_attn(q.to(dev), k.to(dev), vt.to(dev), out, nimg, heads, ntok, pad)
    s = (q16.double() @ k16.double().transpose(-1, -2)) / ops.LOG2E
    ref = (torch.softmax(s, -1) @ v16.double()).permute(0, 2, 1, 3).reshape(nimg * ntok, D)
    err = rel(out.float(), ref)
    assert err < (1e-2 if dtype == torch.bfloat16 else 2e-3), err
    # element-wise as well (a single wrong tile hides in a Frobenius norm) and run-to-run identical (hand-counted waits:
    # a race would come and go)
    emax = float((out.float().cpu().double() - ref).abs().max())
    assert emax < (6e-2 if dtype == torch.bfloat16 else 8e-3), emax
    out2 = torch.zeros_like(out)
    for _ in range(2):
        ops.flash_attn(q.to(dev), k.to(dev), vt.to(dev), out2, nimg, heads, ntok, pad)
        assert torch.equal(out, out2)



@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
def test_flash_attention_extreme_logits(dtype, mode):
    """Logits of several tens and more, as released ViT weights produce in some heads (tests/test_outliers_gpu.py): per query
    a constant offset c_q (softmax does not see it, the kernel's running maximum does) + key-dependent terms, so that
      * queries 0-63: every key of the FIRST tile at <= -150 in the log2 domain (2^-m of the first re-base overflows fp32:
        the accumulators must not be scaled by it), later tiles 200 higher (the re-base path, sums that overflow to +inf);
      * queries 64-127: the first tile carries the row maximum (+160), everything later is 2^-50 of it and less;
      * the rest: ordinary rows with spread 30.
    Every output finite and element-wise equal to an fp64 softmax to the operand type's rounding."""
    from mickey_amd import ops
    dev = _dev()
    ops.attn_set_mode(mode)
    nimg, heads, ntok = 2, 2, 700
    pad = (ntok + 63) // 64 * 64
    gg = g(4242)
    # q = [a_q, 1, c_q, noise ...], k = [1, b_k, d_k, noise ...]: q.k (log2 domain) = a_q + b_k + c_q d_k + small
    qn = torch.randn((nimg, heads, ntok, 64), generator=gg) * 0.5
    kn = torch.randn((nimg, heads, ntok, 64), generator=gg) * 0.5
    a, b, c, d = torch.zeros(ntok), torch.zeros(ntok), torch.zeros(ntok), torch.zeros(ntok)
    b[:64] = -90.0
    b[64:] = 10.0 * torch.randn(ntok - 64, generator=gg).clamp(-3, 3)
    b[300] = 110.0
    a[:64] = -60.0                      # queries 0-63: first tile at -150, key 300 at +50
    c[64:128] = 1.0                     # queries 64-127: first tile at +160, the rest at 0 +- 30, key 300 at +110
    d[:64] = 250.0
    qn[..., 0], qn[..., 1], qn[..., 2] = a, 1.0, c
    kn[..., 0], kn[..., 1], kn[..., 2] = 1.0, b, d
    q2, k16 = qn.to(dtype), kn.to(dtype)
    v16 = torch.randn((nimg, heads, ntok, 64), generator=gg).to(dtype)
    q = torch.zeros((nimg, heads, pad, 64), dtype=dtype)
    k = torch.zeros_like(q)
    vt = torch.zeros((nimg, heads, 64, pad), dtype=dtype)
    q[:, :, :ntok], k[:, :, :ntok] = q2, k16
    t = torch.arange(ntok)
    perm = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    vt[:, :, :, perm] = v16.transpose(-1, -2)
    out = torch.zeros((nimg * ntok, heads * 64), device=dev, dtype=dtype)
    ops.flash_attn(q.to(dev), k.to(dev), vt.to(dev), out, nimg, heads, ntok, pad)
    s2 = q2.double() @ k16.double().transpose(-1, -2)                    # log2 domain
    assert float(s2[..., :64, :64].max()) < -128.0 and float(s2[..., 64:128, :64].min()) > 140.0 and float(s2[..., 64:128, 64:].max()) < 125.0
    ref = (torch.softmax(s2 * 0.6931471805599453, -1) @ v16.double()).permute(0, 2, 1, 3).reshape(nimg * ntok, heads * 64)
    o = out.float().cpu().double()
    assert torch.isfinite(o).all()
    emax = float((o - ref).abs().max())
    assert emax < (6e-2 if dtype == torch.bfloat16 else 8e-3), emax
    ops.attn_set_mode(0)


@pytest.mark.parametrize("pair", [(1, 2), (4, 5)])
def test_flash_attention_queries_per_wave_bit_identical(pair):
    """32 and 64 queries per wave of one kernel family (32x32x16: modes 1 / 2, 16x16x32: modes 4 / 5) perform the same
    operations per query in the same order: bit-identical outputs -- what lets the automatic choice depend on the grid size
    (one pair vs a batch) without a pair's result depending on its batch."""
    from mickey_amd import ops
    dev = _dev()
    nimg, heads, ntok, pad = 3, 4, 1939, 1984
    gq = torch.Generator(device="cpu").manual_seed(5)
    q = (torch.randn((nimg, heads, pad, 64), generator=gq) * 0.3).bfloat16().to(dev)
    k = torch.randn((nimg, heads, pad, 64), generator=gq).bfloat16().to(dev)
    vt = torch.randn((nimg, heads, 64, pad), generator=gq).bfloat16().to(dev)
    outs = []
    for mode in pair:
        ops.attn_set_mode(mode)
        out = torch.zeros((nimg * ntok, heads * 64), device=dev, dtype=torch.bfloat16)
        ops.flash_attn(q, k, vt, out, nimg, heads, ntok, pad)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


def test_split_planes_saturation_is_reported():
    """x * 64 = hi + lo in fp16 saturates at |x| > 1023.  With a watcher word passed to the call (mickey_hip.h: sat_flag) every
    plane-writing kernel reports a clamp; in range nothing is reported; without a word nothing is touched.  A NaN is reported
    AND stays a NaN in both planes (it must reach the outputs' finite checks, not turn into -65504)."""
    from mickey_amd import ops
    dev = _dev()
    flag = torch.zeros((1,), device=dev, dtype=torch.int32)
    x = torch.randn((64, 256), generator=g(3)).to(dev) * 100.0          # |x| < 1023
    hi, lo = ops.plane_pair(x.shape, dev)
    w, b = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    ops.split_planes(x, hi, lo, sat=flag)
    assert int(flag.item()) == 0
    assert rel((hi.float() + lo.float()) / 64.0, x) < 1e-6
    x2 = x.clone()
    x2[5, 7] = 2000.0                                                # beyond the planes' range
    x2[6, 8] = -5000.0
    ops.split_planes(x2, hi, lo, sat=flag)
    assert int(flag.item()) == 1 and float(hi[5, 7]) == 65504.0 and float(hi[6, 8]) == -65504.0
    flag.zero_()
    x3 = x.clone()
    x3[9, 1] = float("nan")
    ops.split_planes(x3, hi, lo, sat=flag)
    assert int(flag.item()) == 1                                     # NaN is reported ...
    assert bool(torch.isnan(hi[9, 1])) and bool(torch.isnan(lo[9, 1]))   # ... and propagated
    assert int(torch.isnan(hi).sum()) == 1
    flag.zero_()
    ops.layernorm(x * 0.01, w * 3000.0, b, 1e-6, out=(hi, lo), sat=flag)   # LayerNorm output (unit variance) x 3000: out of range
    assert int(flag.item()) == 1
    flag.zero_()
    ops.split_planes(x2, hi, lo)
    assert int(flag.item()) == 0                                     # nobody watches: the word is left alone


def test_saturation_words_of_two_streams_are_independent():
    """The saturation word is a per-call argument (no process-wide watcher): two plane-writing pipelines interleaved on two
    streams, one fed out-of-range values and one in range, report into their own words only -- split_planes, the LayerNorm
    that writes planes, and the plane-writing epilogues of the split conv / split grouped GEMM."""
    from mickey_amd import ops, weights
    dev = _dev()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    fa = torch.zeros((1,), device=dev, dtype=torch.int32)
    fb = torch.zeros((1,), device=dev, dtype=torch.int32)
    G, M, N, K = 2, 3000, 128, 128
    a_ok = torch.randn((G, M, K), generator=g(1)).to(dev)
    a_big = a_ok * 40.0                                   # products of ~ +-40 * sqrt(K): the OUTPUT planes (x 64) overflow
    w = torch.randn((G, N, K), generator=g(2))
    ws = weights.split_conv_weight(w).to(dev).contiguous()
    lnw, lnb = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    torch.cuda.synchronize()
    outs = {}
    for it in range(3):
        for name, st, flag, src in (("a", sa, fa, a_big), ("b", sb, fb, a_ok)):
            with torch.cuda.stream(st):
                ah, al = ops.plane_pair(src.shape, dev)
                ops.split_planes(src, ah, al, sat=flag)
                oh, ol = ops.plane_pair((G, M, N), dev)
                ops.gemm_grouped_split((ah, al), ws, None, (oh, ol), G, M, N, K, K, N, M * K, N * 2 * K, 0, M * N, sat=flag)
                lh, ll = ops.plane_pair((G * M, K), dev)
                ops.layernorm(src.reshape(G * M, K), lnw * (2000.0 if name == "a" else 1.0), lnb, 1e-6, out=(lh, ll), sat=flag)
                outs[name] = (oh, ol)
    torch.cuda.synchronize()
    assert int(fa.item()) == 1 and int(fb.item()) == 0
    assert bool(torch.isfinite(outs["b"][0]).all())


def _to_bordered(x, nimg, H, W):
    """dense NHWC rows [..., nimg*H*W, C] -> bordered feature map [..., bordered_rows, C] (zeros elsewhere)"""
    from mickey_amd import ops
    out = ops.bordered_empty(x.shape[:-2], nimg, H, W, x.shape[-1], x.dtype, x.device)
    out[..., ops.bordered_index(nimg, H, W, x.device), :] = x
    return out


@pytest.mark.parametrize("tile", [1, 2, 7])
@pytest.mark.parametrize("with_sc,with_res", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("out_kind", ["f32", "dense", "bordered"])
def test_conv3x3(with_sc, with_res, tile, out_kind):
    from mickey_amd import ops
    dev = _dev()
    ops.gemm_set_tile(tile)
    G, nimg, H, W, C1, C2, Cout = 3, 2, 9, 7, 128, 64, 128 if with_res else 192
    if with_res:
        Cout = C1
    M = nimg * H * W
    x = torch.randn((G, nimg, H, W, C1), generator=g(1)).bfloat16()
    x2 = torch.randn((nimg, H, W, C2), generator=g(2)).bfloat16()
    wc = (torch.randn((G, Cout, C1, 3, 3), generator=g(3)) / math.sqrt(9 * C1)).bfloat16()
    ws = (torch.randn((G, Cout, C2), generator=g(4)) / math.sqrt(C2)).bfloat16()
    bias = torch.randn((G, Cout), generator=g(5))
    K = 9 * C1 + (C2 if with_sc else 0)
    w2d = torch.zeros((G, Cout, K), dtype=torch.bfloat16)
    w2d[:, :, : 9 * C1] = wc.permute(0, 1, 3, 4, 2).reshape(G, Cout, 9 * C1)
    if with_sc:
        w2d[:, :, 9 * C1:] = ws
    R = ops.bordered_rows(nimg, H, W)
    xd = _to_bordered(x.reshape(G, M, C1).to(dev), nimg, H, W)          # [G, R, C1]
    x2d = _to_bordered(x2.reshape(M, C2).to(dev), nimg, H, W)           # [R, C2], shared by the groups
    if out_kind == "f32":
        out = torch.empty((G, M, Cout), device=dev, dtype=torch.float32)
    elif out_kind == "dense":
        out = torch.empty((G, M, Cout), device=dev, dtype=torch.bfloat16)
    else:   # border rows pre-filled with a sentinel: the kernel must not touch them
        out = torch.full((G, R, Cout), 7.0, device=dev, dtype=torch.bfloat16)
    ops.conv3x3(xd, C1, w2d.to(dev), bias.to(dev), out, Cout, G, nimg, H, W, act=ops.ACT_RELU,
                in2=x2d if with_sc else None, C2=C2, resid=xd if with_res else None,
                stride_in1=R * C1, stride_in2=0, stride_w=Cout * K, stride_bias=Cout,
                stride_out=out.shape[1] * Cout, stride_resid=R * C1, out_bordered=out_kind == "bordered")
    if out_kind == "bordered":
        idx = ops.bordered_index(nimg, H, W, dev)
        mask = torch.ones(R, dtype=torch.bool, device=dev)
        mask[idx] = False
        assert bool((out[:, mask] == 7.0).all())
        out = out[:, idx]
    for gi in range(G):
        ref = F.conv2d(x[gi].float().permute(0, 3, 1, 2), wc[gi].float(), padding=1) + bias[gi].view(1, -1, 1, 1)
        if with_sc:
            ref = ref + F.conv2d(x2.float().permute(0, 3, 1, 2), ws[gi].float()[:, :, None, None])
        if with_res:
            ref = ref + x[gi].float().permute(0, 3, 1, 2)
        ref = F.relu(ref).permute(0, 2, 3, 1).reshape(-1, Cout)
        tol = 2e-5 if out_kind == "f32" else 3e-3   # 16-bit outputs: one bf16 rounding
        assert rel(out[gi].float(), ref) < tol, (gi, rel(out[gi].float(), ref))


@pytest.mark.parametrize("nimg,H,W", [(64, 51, 38), (3, 5, 3)])
def test_conv3x3_big_tiles_and_tiny_grids(nimg, H, W):
    """the automatic schedule at the bench shape (256x256 ping-pong tiles: interior + edge tiles, rows walked in the LDS
    epilogue across many image boundaries) and a grid narrower than the epilogue's row step (W < 8)"""
    from mickey_amd import ops
    dev = _dev()
    ops.gemm_set_tile(0)
    C1, Cout = 64, 256
    M = nimg * H * W
    x = torch.randn((M, C1), generator=g(1)).bfloat16().to(dev)
    wc = (torch.randn((Cout, C1, 3, 3), generator=g(3)) / math.sqrt(9 * C1)).bfloat16()
    w2d = wc.permute(0, 2, 3, 1).reshape(Cout, 9 * C1).contiguous().to(dev)
    bias = torch.randn((Cout,), generator=g(5)).to(dev)
    R = ops.bordered_rows(nimg, H, W)
    xd = _to_bordered(x, nimg, H, W)
    out = torch.full((R, Cout), 7.0, device=dev, dtype=torch.bfloat16)
    ops.conv3x3(xd, C1, w2d, bias, out, Cout, 1, nimg, H, W, act=ops.ACT_NONE, out_bordered=True)
    idx = ops.bordered_index(nimg, H, W, dev)
    mask = torch.ones(R, dtype=torch.bool, device=dev)
    mask[idx] = False
    assert bool((out[mask] == 7.0).all())
    ref = F.conv2d(x.float().reshape(nimg, H, W, C1).permute(0, 3, 1, 2), wc.float().to(dev), padding=1) + bias.view(1, -1, 1, 1)
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    assert rel(out[idx].float(), ref) < 3e-3
    # and the chain: a second conv reading what the first one wrote (borders intact) == conv of the dense rows
    out2 = torch.empty((M, Cout), device=dev, dtype=torch.float32)
    out[mask] = 0
    wc2 = (torch.randn((Cout, Cout, 3, 3), generator=g(7)) / math.sqrt(9 * Cout)).bfloat16()
    ops.conv3x3(out, Cout, wc2.permute(0, 2, 3, 1).reshape(Cout, 9 * Cout).contiguous().to(dev), bias, out2, Cout, 1, nimg, H, W)
    ref2 = F.conv2d(out[idx].float().reshape(nimg, H, W, Cout).permute(0, 3, 1, 2), wc2.float().to(dev), padding=1) + bias.view(1, -1, 1, 1)
    assert rel(out2, ref2.permute(0, 2, 3, 1).reshape(M, Cout)) < 2e-5


def _zero_lo_pair(xh, ops, dev):
    """(hi, lo) planes in ONE allocation (what the split kernels address) with lo identically zero"""
    h, l = ops.plane_pair(xh.shape, dev)
    h.copy_(xh)
    l.zero_()
    return h, l


@pytest.mark.parametrize("nimg,H,W,G,C1,C2,Cout,bordered", [(2, 9, 7, 3, 128, 64, 192, True),     # 128x128 tiles, shortcut source
                                                            (2, 9, 7, 2, 64, 0, 64, False),       # no second source, dense rows
                                                            (2, 9, 7, 2, 96, 32, 64, True),       # channel counts of 32-granularity
                                                            (32, 51, 38, 1, 64, 64, 256, True),   # 256x256 ping-pong tiles
                                                            (32, 51, 38, 1, 96, 32, 256, False),  # ... at 32-granularity (odd stage counts)
                                                            (32, 51, 38, 2, 64, 0, 320, False)])
def test_conv3x3_split_is_fp32_grade(nimg, H, W, G, C1, C2, Cout, bordered):
    """mk_conv3x3_split: fp32 activations / weights as fp16 hi + lo planes staged once, three MFMA products -- against an fp64 evaluation of
    the SAME fp32 values its error must be that of an fp32 evaluation (beside it), far below one fp16 rounding (2^-11 =
    4.9e-4).  Activations carry a wide range of magnitudes (the lo plane's subnormal end) and exact zeros (post-ReLU); both
    kernels (128x128 and the 256x256 ping-pong), shared / per-group sources, dense and bordered fp32 output.  The references
    are im2col matmuls on the GPU (fp64 / fp32)."""
    from mickey_amd import ops, weights
    dev = _dev()
    ops.gemm_set_tile(0)
    M = nimg * H * W
    x = torch.randn((G, M, C1), generator=g(1)) * torch.exp(1.5 * torch.randn((G, M, 1), generator=g(11)))
    x = F.relu(x).clamp_(max=900.0).to(dev)
    x2 = (torch.randn((M, C2), generator=g(2)) * 3.0).to(dev) if C2 else None
    wc = torch.randn((G, Cout, C1, 3, 3), generator=g(3)) / math.sqrt(9 * C1)
    wsc = torch.randn((G, Cout, C2), generator=g(4)) / math.sqrt(C2) if C2 else None
    bias = torch.randn((G, Cout), generator=g(5)).to(dev)
    w2d = wc.permute(0, 1, 3, 4, 2).reshape(G, Cout, 9 * C1)
    if C2:
        w2d = torch.cat([w2d, wsc], 2)
    K = w2d.shape[2]
    wsplit = weights.split_conv_weight(w2d).to(dev).contiguous()
    assert wsplit.shape == (G, Cout, 2 * K) and wsplit.dtype == torch.float16
    R = ops.bordered_rows(nimg, H, W)
    xb = _to_bordered(x, nimg, H, W)                         # fp32 [G, R, C1]
    xh, xl = ops.split_planes(xb, *ops.plane_pair(xb.shape, dev))
    resid = (xh.double() + xl.double() - xb.double() * ops.SPLIT_ACT_SCALE).abs().max()
    assert float(resid) <= 2.0 ** -21 * float(xb.max()) * ops.SPLIT_ACT_SCALE + 1e-7, float(resid)
    in2 = None
    if C2:
        x2b = _to_bordered(x2, nimg, H, W)
        in2 = ops.split_planes(x2b, *ops.plane_pair(x2b.shape, dev))
    out = torch.full((G, R if bordered else M, Cout), 7.0, device=dev, dtype=torch.float32)
    out_raw = out
    ops.conv3x3_split((xh, xl), C1, wsplit, bias, out, Cout, G, nimg, H, W, act=ops.ACT_RELU, in2=in2, C2=C2,
                      stride_in1=R * C1, stride_in2=0, stride_w=Cout * 2 * K, stride_bias=Cout, stride_out=out.shape[1] * Cout,
                      out_bordered=bordered)
    if bordered:
        idx = ops.bordered_index(nimg, H, W, dev)
        mask = torch.ones(R, dtype=torch.bool, device=dev)
        mask[idx] = False
        assert bool((out[:, mask] == 7.0).all())
        out = out[:, idx]
    worst = 0.0
    for gi in range(G):
        xp = F.pad(x[gi].reshape(nimg, H, W, C1), (0, 0, 1, 1, 1, 1))
        cols = torch.cat([xp[:, ky:ky + H, kx:kx + W].reshape(M, C1) for ky in range(3) for kx in range(3)] + ([x2] if C2 else []), 1)
        wg = w2d[gi].to(dev)

        def ref(dt):
            return F.relu(cols.to(dt) @ wg.to(dt).t() + bias[gi].to(dt))
        ref64 = ref(torch.float64)
        e_split, e_f32 = rel(out[gi], ref64), rel(ref(torch.float32), ref64)
        worst = max(worst, e_split)
        assert e_split < 2e-6 and e_split < 4 * e_f32 + 2e-7, (gi, e_split, e_f32)
    print("split conv vs fp64: %.2e" % worst)
    # the same conv writing the NEXT split conv's operand planes instead of fp32 rows: (hi + lo) / scale == the fp32 result to
    # the planes' 22 bits, border rows untouched
    rows = R if bordered else M
    ph = torch.full((G, rows, Cout), 7.0, device=dev, dtype=torch.float16)
    pl = torch.full((G, rows, Cout), 7.0, device=dev, dtype=torch.float16)
    ops.conv3x3_split((xh, xl), C1, wsplit, bias, (ph, pl), Cout, G, nimg, H, W, act=ops.ACT_RELU, in2=in2, C2=C2,
                      stride_in1=R * C1, stride_in2=0, stride_w=Cout * 2 * K, stride_bias=Cout, stride_out=rows * Cout,
                      out_bordered=bordered)
    if bordered:
        assert bool((ph[:, mask] == 7.0).all()) and bool((pl[:, mask] == 7.0).all())
        ph, pl = ph[:, idx], pl[:, idx]
    back = (ph.double() + pl.double()) / ops.SPLIT_ACT_SCALE
    assert float((back - out.double()).abs().max()) <= 2.0 ** -21 * float(out.abs().max()) + 1e-7
    # in1 = (hi, None): the source IS its hi plane (fp16 features of an fp16 encoder, AMD.FEATURES_LP): two products instead of
    # three -- fp32-grade against an fp64 evaluation of those fp16 values, and equal to the three-product kernel fed a zero lo plane
    if not C2:
        out2 = torch.full_like(out_raw, 7.0)
        ops.conv3x3_split((xh, None), C1, wsplit, bias, out2, Cout, G, nimg, H, W, act=ops.ACT_RELU, stride_in1=R * C1,
                          stride_w=Cout * 2 * K, stride_bias=Cout, stride_out=out2.shape[1] * Cout, out_bordered=bordered)
        out3 = torch.full_like(out_raw, 7.0)
        ops.conv3x3_split(_zero_lo_pair(xh, ops, dev), C1, wsplit, bias, out3, Cout, G, nimg, H, W,
                          act=ops.ACT_RELU, stride_in1=R * C1, stride_w=Cout * 2 * K, stride_bias=Cout, stride_out=out3.shape[1] * Cout,
                          out_bordered=bordered)
        assert torch.equal(out2, out3)
        o2 = out2[:, idx] if bordered else out2
        xhi = (xh.float() / ops.SPLIT_ACT_SCALE)[:, ops.bordered_index(nimg, H, W, dev)]      # the fp16 values, as dense rows
        for gi in range(G):
            xp = F.pad(xhi[gi].reshape(nimg, H, W, C1), (0, 0, 1, 1, 1, 1))
            cols = torch.cat([xp[:, ky:ky + H, kx:kx + W].reshape(M, C1) for ky in range(3) for kx in range(3)], 1)
            ref64 = F.relu(cols.double() @ w2d[gi].to(dev).double().t() + bias[gi].double())
            assert rel(o2[gi], ref64) < 2e-6, (gi, rel(o2[gi], ref64))


@pytest.mark.parametrize("M,N,K,lda", [(500, 128, 128, 256), (40000, 384, 128, 256), (40000, 256, 256, 256), (3000, 128, 256, 256),
                                       (40000, 256, 96, 256), (700, 64, 32, 64)])
def test_grouped_gemm_split_is_fp32_grade(M, N, K, lda):
    """mk_gemm_grouped_split (the heads' small linears in AMD.HEADS_DTYPE: split): fp32 A [G, M, lda] as (hi, lo) planes -- also
    converted as two column blocks at different times, as the pipeline does with `cat` -- times fp32 weights in the
    interleaved (32 hi | 32 lo) layout, both kernels (128x128; 256x256 ping-pong for the 40000-row cases), fp32 output and the
    plane-output form with ReLU; against fp64."""
    from mickey_amd import ops, weights
    dev = _dev()
    ops.gemm_set_tile(0)
    G = 4
    a = (torch.randn((G, M, lda), generator=g(1)) * torch.exp(torch.randn((G, M, 1), generator=g(2)))).to(dev)
    w = (torch.randn((G, N, K), generator=g(3)) / math.sqrt(K))
    ws = weights.split_conv_weight(w).to(dev).contiguous()
    ah, al = ops.plane_pair(a.shape, dev)
    half = lda // 2
    ops.split_planes(a[:, :, :half], ah[:, :, :half], al[:, :, :half])
    ops.split_planes(a[:, :, half:], ah[:, :, half:], al[:, :, half:])
    assert float((ah.double() + al.double() - a.double() * ops.SPLIT_ACT_SCALE).abs().max()) <= 2.0 ** -21 * float(a.abs().max()) * 64 + 1e-7
    out = torch.empty((G, M, N), device=dev, dtype=torch.float32)
    ops.gemm_grouped_split((ah, al), ws, None, out, G, M, N, K, lda, N, M * lda, N * 2 * K, 0, M * N)
    ref64 = torch.einsum("gmk,gnk->gmn", a[:, :, :K].double(), w.to(dev).double())
    ref32 = torch.einsum("gmk,gnk->gmn", a[:, :, :K], w.to(dev))
    e, e32 = rel(out, ref64), rel(ref32, ref64)
    print("split grouped GEMM vs fp64 %.2e (torch fp32 %.2e)" % (e, e32))
    assert e < 2e-6 and e < 4 * e32 + 2e-7
    oh, ol = torch.empty((G, M, N), device=dev, dtype=torch.float16), torch.empty((G, M, N), device=dev, dtype=torch.float16)
    ops.gemm_grouped_split((ah, al), ws, None, (oh, ol), G, M, N, K, lda, N, M * lda, N * 2 * K, 0, M * N, act=ops.ACT_RELU)
    back = (oh.double() + ol.double()) / ops.SPLIT_ACT_SCALE
    assert float((back - F.relu(out).double()).abs().max()) <= 2.0 ** -21 * float(out.abs().max()) + 1e-7


def test_grouped_gemm():
    from mickey_amd import ops
    dev = _dev()
    G, M, N, K = 4, 500, 384, 128
    a = torch.randn((G, M, 256), generator=g(1)).bfloat16()   # lda 256 > K
    w = (torch.randn((G, N, K), generator=g(2)) / math.sqrt(K)).bfloat16()
    out = torch.empty((G, M, N), device=dev, dtype=torch.float32)
    ops.gemm_grouped(a.to(dev), w.to(dev), None, out, G, M, N, K, 256, K, N, M * 256, N * K, 0, M * N)
    ref = torch.einsum("gmk,gnk->gmn", a[:, :, :K].float(), w.float())
    assert rel(out, ref) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K,with_resid,bordered,nimg,H,W", [(128, False, False, 2, 9, 7), (256, True, False, 2, 9, 7), (256, True, True, 2, 9, 7),
                                                            (128, False, False, 64, 51, 38), (256, True, True, 64, 51, 38)])
def test_gemm_ln128_equals_gemm_then_layernorm(K, with_resid, bordered, nimg, H, W, dtype):
    """mk_gemm_ln128 (round 6: Linear(K -> 128, no bias) + LayerNorm (+ residual) in one pass, the fp32 GEMM output never written)
    against the two launches it replaces in the heads' linear-attention layers -- mk_gemm_grouped (fp32 out) then mk_layernorm with
    per-group weights: the accumulators are the same (K steps in order), the row statistics are summed in another order -- fp32
    round-off on the updated residual, at most one ulp of the 16-bit type on the written rows -- and against fp64; 4 groups, rows not
    a multiple of the tile (126), dense and bordered output, the bench geometry (4 x 124032 rows)."""
    from mickey_amd import ops
    dev = _dev()
    G, C = 4, 128
    M = nimg * H * W
    gen = torch.Generator(device="cuda").manual_seed(31)
    rn = lambda *shape, s=1.0: torch.randn(shape, device=dev, generator=gen) * s  # noqa: E731
    a = rn(G, M, K, s=0.7).to(dtype)
    w = rn(G, C, K, s=1.5 / math.sqrt(K)).to(dtype)
    lw, lb = 1.0 + 0.3 * rn(G, C), 0.2 * rn(G, C)
    res0 = rn(G, M, C) if with_resid else None
    R = ops.bordered_rows(nimg, H, W)
    ldo = C if bordered else 2 * C

    def outbuf():
        return torch.full((G, R if bordered else M, ldo), 7.0, device=dev, dtype=dtype)
    # the two launches
    mrg = torch.empty((G, M, C), device=dev, dtype=torch.float32)
    ops.gemm_grouped(a, w, None, mrg, G, M, C, K, K, K, C, M * K, C * K, 0, M * C)
    res_a = res0.clone() if with_resid else None
    out_a = outbuf()
    ops.layernorm(mrg, lw, lb, 1e-5, out=out_a, ldo=ldo, resid=res_a, rows_out=G * M, rows_per_img=G * M, wgroup_rows=M,
                  bordered=(nimg, H, W) if bordered else None)
    # one launch
    res_b = res0.clone() if with_resid else None
    out_b = outbuf()
    ops.gemm_ln128(a, w, lw, lb, 1e-5, out_b, G, M, K, ldo=ldo, resid=res_b, bordered=(nimg, H, W) if bordered else None)
    if bordered:
        idx = ops.bordered_index(nimg, H, W, dev)
        mask = torch.ones(R, dtype=torch.bool, device=dev)
        mask[idx] = False
        assert bool((out_b[:, mask] == 7.0).all())                       # border rows untouched
        out_a, out_b = out_a[:, idx], out_b[:, idx]
    else:
        assert bool((out_b[:, :, C:] == 7.0).all())                      # the other half of the 2C-wide rows untouched
        out_a, out_b = out_a[:, :, :C], out_b[:, :, :C]
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    d = (out_a.float() - out_b.float()).abs() / out_a.float().abs().clamp_min(1e-3)
    assert float(d.max()) <= 1.01 * ulp and float((d > 0).float().mean()) < 0.02, (float(d.max()), float((d > 0).float().mean()))
    if with_resid:
        assert rel(res_b, res_a) < 2e-7
    # fp64
    y = torch.einsum("gmk,gnk->gmn", a.double(), w.double())
    y = (y - y.mean(-1, keepdim=True)) / torch.sqrt(y.var(-1, unbiased=False, keepdim=True) + 1e-5) * lw.double()[:, None] + lb.double()[:, None]
    if with_resid:
        y = y + res0.double()
        assert rel(res_b, y) < 2e-6
    assert rel(out_b.float(), y) < (6e-4 if dtype == torch.float16 else 5e-3)


def test_linear_attention_and_posenc():
    from mickey_amd import ops
    from oracle import mickey_oracle as O
    dev = _dev()
    G, nimg, h, w, C = 2, 2, 9, 7, 128
    L = h * w
    qkv = torch.randn((G, nimg * L, 3 * C), generator=g(1))
    qd = qkv.to(dev)
    kv = torch.empty((G * nimg * (C // 16), 272), device=dev)
    work = torch.empty((ops.linattn_work_floats(G, nimg, L, C),), device=dev)
    ops.linattn_kv(qd, kv, work, G, nimg, L, C)
    out = torch.empty((G, nimg * L, C), device=dev, dtype=torch.bfloat16)
    ops.linattn_apply(qd, kv, out, C, G, nimg, L, C)
    x = qkv.reshape(G * nimg, L, 3, 8, 16)
    ref = O.linear_attention(x[:, :, 0], x[:, :, 1], x[:, :, 2]).reshape(G, nimg * L, C)
    assert rel(out.float(), ref) < 4e-3
    # positional encoding add
    xin = torch.randn((G, nimg * L, C), generator=g(2)).bfloat16()
    pe = O.sine_pos_encoding(C, h, w).reshape(C, L).t().contiguous()
    xs = torch.empty((G, nimg * L, C), device=dev)
    cat = torch.zeros((G, nimg * L, 2 * C), device=dev, dtype=torch.bfloat16)
    ops.posenc_add(xin.to(dev), pe.to(dev), xs, cat, G, nimg, L, C)
    refx = xin.float() + pe.repeat(nimg, 1)[None]
    assert rel(xs, refx) < 1e-6 and rel(cat[:, :, :C].float(), refx) < 4e-3


def test_head_tails():
    from mickey_amd import ops
    from oracle import mickey_oracle as O
    dev = _dev()
    nimg, h, w, C, Cd = 2, 11, 9, 64, 128
    n = h * w
    fd, fo, fz = (torch.randn((nimg * n, C), generator=g(i)) for i in (1, 2, 3))
    fdsc = torch.randn((nimg * n, Cd), generator=g(4))
    ws, wxy, wd = torch.randn((C,), generator=g(5)) * 4, torch.randn((2, C), generator=g(6)) * 0.3, torch.randn((C,), generator=g(7))
    scr, kps, depth, dsc = ops.head_tails(fd.to(dev), ws.to(dev), fo.to(dev), wxy.to(dev), fz.to(dev), wd.to(dev), fdsc.to(dev),
                                          nimg, h, w, C, Cd)
    def nchw(f, c):
        return f.reshape(nimg, h, w, c).permute(0, 3, 1, 2)
    s_ref = O.border_softmax(F.conv2d(nchw(fd, C), ws.view(1, C, 1, 1)), 3).reshape(nimg, 1, n)
    k_ref = O.abs_keypoints(torch.sigmoid(F.conv2d(nchw(fo, C), wxy.view(2, C, 1, 1))), 14).reshape(nimg, 2, n)
    d_ref = F.conv2d(nchw(fz, C), wd.view(1, C, 1, 1)).reshape(nimg, 1, n)
    x = nchw(fdsc, Cd)
    ds_ref = (x / x.pow(2).sum(1, keepdim=True).add(1e-10).pow(0.5)).reshape(nimg, Cd, n)
    assert rel(scr, s_ref) < 1e-5 and rel(kps, k_ref) < 1e-6 and rel(depth, d_ref) < 1e-5 and rel(dsc, ds_ref) < 1e-6
    assert abs(float(scr.sum()) - nimg) < 1e-4


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("n0,n1", [(150, 131), (1938, 1938)])
def test_dual_softmax_vs_oracle(n0, n1, split):
    """split = False: the exact fp32 MFMA correlation; True: split-fp16 operands on the 16-bit matrix cores (mk_dual_softmax_split)."""
    from mickey_amd import ops
    from oracle import mickey_oracle as O
    dev = _dev()
    B = 2
    d0 = F.normalize(torch.randn((B, 128, n0), generator=g(21)), dim=1)
    d1 = F.normalize(torch.randn((B, 128, n1), generator=g(22)) + 0.5 * d0[:, :, :n1] if n1 <= n0 else
                     torch.randn((B, 128, n1), generator=g(22)), dim=1)
    s0 = torch.rand((B, 1, n0), generator=g(23)) / n0
    s1 = torch.rand((B, 1, n1), generator=g(24)) / n1
    ref = O.dual_softmax(d0, d1, 0.7, 0.1)
    kp_ref = torch.matmul(s0.transpose(2, 1), s1)
    sc, kp, fin = ops.dual_softmax(d0.to(dev), d1.to(dev), s0.to(dev), s1.to(dev), 0.1, 0.7, split=split)
    assert rel(sc, ref) < 1e-5, rel(sc, ref)
    assert torch.equal(kp.cpu(), kp_ref)
    assert rel(fin, ref * kp_ref) < 1e-5
    # row arg-max indices identical wherever the oracle's own top-2 gap is not a tie
    top2 = ref.topk(2, dim=2).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-5 * top2[..., 0]
    assert torch.equal(sc.cpu().argmax(2)[clear], ref.argmax(2)[clear])
    # the same for the column arg-max (the other direction of the mutual-NN check)
    top2c = ref.topk(2, dim=1).values
    clearc = (top2c[:, 0] - top2c[:, 1]) > 1e-5 * top2c[:, 0]
    assert torch.equal(sc.cpu().argmax(1)[clearc], ref.argmax(1)[clearc])
    # no dustbin
    sc2, _, _ = ops.dual_softmax(d0.to(dev), d1.to(dev), None, None, 0.1, None, want_kp=False, want_final=False, split=split)
    assert rel(sc2, O.dual_softmax(d0, d1, None, 0.1)) < 1e-5
    # lean call (final_scores only) == the full call's final_scores, bit for bit
    _, _, fin2 = ops.dual_softmax(d0.to(dev), d1.to(dev), s0.to(dev), s1.to(dev), 0.1, 0.7, want_scores=False, want_kp=False, split=split)
    assert torch.equal(fin2, fin)


def test_split_fp16_correlation_is_fp32_grade():
    """The split-operand correlation against an fp64 evaluation of the dual softmax, next to the exact-fp32 MFMA path and the
    oracle's own fp32 matmul: its error must be of the SAME order (the three share the fp32 round-off of the exponent, ~|v2| 6e-8),
    on unit-norm descriptors with planted near-duplicates (large logits) and with many tiny entries (the lo plane's range)."""
    from mickey_amd import ops
    from oracle import mickey_oracle as O
    dev = _dev()
    B, n = 2, 700
    d0 = torch.randn((B, 128, n), generator=g(41))
    d0[:, 64:] *= 1e-3                                   # half of the channels tiny: hi underflows fp16's normal range unscaled
    d0 = F.normalize(d0, dim=1)
    d1 = F.normalize(d0 + 0.05 * torch.randn((B, 128, n), generator=g(42)), dim=1)    # every keypoint has a near-duplicate: S ~ 1
    ref64 = O.dual_softmax(d0.double(), d1.double(), 1.0, 0.1)
    errs = {}
    for name, kw in (("exact", dict(split=False)), ("split", dict(split=True))):
        sc, _, _ = ops.dual_softmax(d0.to(dev), d1.to(dev), None, None, 0.1, 1.0, want_kp=False, want_final=False, **kw)
        errs[name] = rel(sc, ref64)
        assert torch.equal(sc.cpu().argmax(2), ref64.argmax(2)) and torch.equal(sc.cpu().argmax(1), ref64.argmax(1))
    errs["oracle_fp32"] = rel(O.dual_softmax(d0, d1, 1.0, 0.1), ref64)
    print("dual softmax vs fp64:", {k: "%.2e" % v for k, v in errs.items()})
    assert errs["split"] < 3e-6 and errs["exact"] < 3e-6
    assert errs["split"] < 3 * max(errs["exact"], errs["oracle_fp32"])


@pytest.mark.parametrize("B,C,n0,n1,split", [(3, 64, 77, 200, False), (9, 128, 97, 64, False), (1, 32, 33, 31, False),
                                            (9, 128, 97, 64, True), (11, 128, 70, 196, True), (2, 128, 33, 31, True)])
def test_dual_softmax_other_shapes(B, C, n0, n1, split):
    """Descriptor widths below 128 (the generic-C instantiation), ragged n0 != n1 not multiples of 32, and B >= 8 (the
    XCD-local workgroup decode) against the oracle; split path: output rows 16-byte aligned (n1 % 4 == 0: 16-byte stores),
    8-byte aligned and odd (single floats), fewer column tiles than a wave's pair."""
    from mickey_amd import ops
    from oracle import mickey_oracle as O
    dev = _dev()
    d0 = F.normalize(torch.randn((B, C, n0), generator=g(31)), dim=1)
    d1 = F.normalize(torch.randn((B, C, n1), generator=g(32)), dim=1)
    s0 = torch.rand((B, 1, n0), generator=g(33)) / n0
    s1 = torch.rand((B, 1, n1), generator=g(34)) / n1
    ref = O.dual_softmax(d0, d1, 1.0, 0.1)
    kp_ref = torch.matmul(s0.transpose(2, 1), s1)
    sc, kp, fin = ops.dual_softmax(d0.to(dev), d1.to(dev), s0.to(dev), s1.to(dev), 0.1, 1.0, split=split)
    assert rel(sc, ref) < 1e-5, rel(sc, ref)
    assert torch.equal(kp.cpu(), kp_ref)
    assert rel(fin, ref * kp_ref) < 1e-5
    if split:   # the dev knob of pass 2 (column chunks per row block) changes the schedule, not a bit of the result
        ops.dual_softmax_set_chunks(3)
        sc2, kp2, fin2 = ops.dual_softmax(d0.to(dev), d1.to(dev), s0.to(dev), s1.to(dev), 0.1, 1.0, split=True)
        ops.dual_softmax_set_chunks(0)
        assert torch.equal(sc2, sc) and torch.equal(fin2, fin) and torch.equal(kp2, kp)


def test_matcher_golden(golden):
    """HIP matcher against the REFERENCE's own outputs (tests/golden/matcher.npz)."""
    from mickey_amd import ops
    dev = _dev()
    gd = golden("matcher")
    gen = g(21)
    d0 = F.normalize(torch.randn((2, 128, 150), generator=gen), dim=1)
    d1 = F.normalize(torch.randn((2, 128, 131), generator=gen) + 0.5 * d0[:, :, :131], dim=1)
    for split in (False, True):
        sc, _, _ = ops.dual_softmax(d0.to(dev), d1.to(dev), None, None, 0.1, 0.7, want_kp=False, want_final=False, split=split)
        assert rel(sc, torch.from_numpy(gd["dual_softmax"])) < 1e-5
    sk = ops.sinkhorn(d0.to(dev), d1.to(dev), 1.3, 10)
    assert rel(sk, torch.from_numpy(gd["sinkhorn"])) < 2e-5
    # mutual-NN: bit-exact indices given identical scores
    m, c = ops.mutual_nn(torch.from_numpy(gd["dual_softmax"])[:1].contiguous().to(dev))
    cnt = int(c[0])
    assert cnt == gd["mnn"].shape[0]
    assert torch.equal(m[0, :cnt].cpu().long(), torch.from_numpy(gd["mnn"]))
