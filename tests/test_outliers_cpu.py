"""CPU: synthetic.plant_outliers() -- the real-weight-shaped activation statistics tests/test_outliers_gpu.py runs the HIP
path on -- does what its docstring says, in the fp32 oracle."""
import torch


def test_planted_statistics_are_what_the_docstring_says(cfg):
    """CPU-side: the outlier weights really produce |x| in the hundreds and logits of tens in the fp32 oracle (so that the
    tests below test what they claim), and differ from the plain weights only in the encoder."""
    import torch.nn.functional as F
    from mickey_amd import synthetic as syn
    from oracle import mickey_oracle as O
    sd, plain = syn.mickey_state_dict(cfg, seed=0, outliers=True), syn.mickey_state_dict(cfg, seed=0)
    changed = [k for k in sd if not torch.equal(sd[k], plain[k])]
    assert changed and all(k.startswith(syn.DINO_PREFIX) for k in changed)
    p = syn.DINO_PREFIX
    img = syn.synthetic_batch(B=1, H=182, W=196, seed=1234)["image0"]
    with torch.no_grad():
        x = O.vit_prepare_tokens(sd, p, img)
        lo, hi = 0.0, 0.0
        for i in range(24):
            bp = p + "blocks.%d." % i
            xn = F.layer_norm(x, (1024,), sd[bp + "norm1.weight"], sd[bp + "norm1.bias"], 1e-6)
            qkv = F.linear(xn, sd[bp + "attn.qkv.weight"], sd[bp + "attn.qkv.bias"]).reshape(1, -1, 3, 16, 64)
            lg = (qkv[:, :, 0].permute(0, 2, 1, 3) @ qkv[:, :, 1].permute(0, 2, 3, 1)) * 0.125
            lo, hi = min(lo, float(lg.min())), max(hi, float(lg.max()))
            x = O.vit_block(sd, bp, x, 16)
    assert float(x.abs().max()) > 500.0 and float(x.std(-1).mean()) > 20.0     # massive activations, row std ~ 29
    assert hi - lo > 80.0 and hi > 40.0                                         # logits of several tens
