"""What synthetic.plant_outliers() does to the encoder's activations (CPU, fp32 oracle, block by block): max |x| of the
residual stream in the outlier / ordinary channels, row std, attention-logit range per block.

    python tools/diag_outliers.py [H W]        (default 182 196; 720 540 takes ~1 min)
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mickey_amd import synthetic as syn  # noqa: E402
from mickey_amd.config import default_cfg  # noqa: E402
from oracle import mickey_oracle as O  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (182, 196)
cfg = default_cfg()
sd = syn.mickey_state_dict(cfg, seed=0, outliers=True)
ch, start = syn.plant_outliers({k: v.clone() for k, v in syn.dinov2_state_dict("vit_large", 0, syn.DINO_PREFIX).items()})
print("outlier channels", ch.tolist(), "from block", start)
p = syn.DINO_PREFIX
img = syn.synthetic_batch(B=1, H=H, W=W, seed=1234)["image0"][:, :, :14 * (H // 14), :14 * (W // 14)]
torch.set_num_threads(os.cpu_count())
with torch.no_grad():
    x = O.vit_prepare_tokens(sd, p, img)
    D = x.shape[-1]
    ordinary = torch.ones(D, dtype=torch.bool)
    ordinary[ch] = False
    for i in range(24):
        bp = p + "blocks.%d." % i
        xn = F.layer_norm(x, (D,), sd[bp + "norm1.weight"], sd[bp + "norm1.bias"], 1e-6)
        qkv = F.linear(xn, sd[bp + "attn.qkv.weight"], sd[bp + "attn.qkv.bias"]).reshape(x.shape[0], x.shape[1], 3, 16, 64)
        q, k = qkv[:, :, 0].permute(0, 2, 1, 3), qkv[:, :, 1].permute(0, 2, 1, 3)
        logits = (q @ k.transpose(-1, -2)) * 0.125
        per_head = logits.amax(dim=(0, 2, 3)) - logits.amin(dim=(0, 2, 3))
        x = O.vit_block(sd, bp, x, 16)
        print("block %2d: |x| outlier max %7.1f  ordinary max %6.2f std %5.2f  row std %6.2f  LN1 out max %6.2f  logits [%7.1f, %7.1f] widest head range %6.1f"
              % (i, float(x[..., ch].abs().max()), float(x[..., ordinary].abs().max()), float(x[..., ordinary].std()),
                 float(x.std(-1).mean()), float(xn.abs().max()), float(logits.min()), float(logits.max()), float(per_head.max())))
