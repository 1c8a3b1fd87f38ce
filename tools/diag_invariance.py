"""Which schedule family breaks bit-equality between a one-pair forward and the same pair inside a 32-pair forward (dev)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops, pipeline, synthetic as syn  # noqa: E402
from mickey_amd.config import default_cfg  # noqa: E402
from mickey_amd.model import MickeyRelativePose  # noqa: E402

dev = torch.device("cuda:0")
cfg = default_cfg()
cfg["AMD"]["ENCODER_DTYPE"] = "bf16"
cfg["AMD"]["GRAPH"] = False
model = MickeyRelativePose(cfg)
model.load_state_dict(syn.mickey_state_dict(cfg, seed=0))
model = model.cuda()
B = 32
batch = {k: v.to(dev) for k, v in syn.synthetic_batch(B=B, H=720, W=540, seed=1234).items()}
W = model.device_weights()


def feats(data, tile, attn):
    ops.gemm_set_tile(tile)
    ops.attn_set_mode(attn)
    im = torch.cat([data["image0"], data["image1"]], 0).contiguous()
    feat, gh, gw = pipeline.encoder_forward(W, pipeline.Workspace(), im)
    scr, kps, depth, dsc = pipeline.heads_forward(W, pipeline.Workspace(), feat, im.shape[0], gh, gw, model.cfg)
    return feat.clone(), kps.clone(), dsc.clone()


big = feats(batch, 0, 0)
one = {k: v[13:14].contiguous() for k, v in batch.items()}
n = big[0].shape[0] // (2 * B)
for tile, attn in ((0, 0), (7, 0), (0, 2), (7, 2)):
    f, kp, ds = feats(one, tile, attn)
    ef = torch.equal(f[:n], big[0][13 * n:14 * n])
    ek = torch.equal(kp[0], big[1][13])
    print("one pair with gemm tile %d attn mode %d: encoder features equal %s, keypoints equal %s (max |d feat| %.3g)" %
          (tile, attn, ef, ek, float((f[:n].float() - big[0][13 * n:14 * n].float()).abs().max())))
ops.gemm_set_tile(0)
ops.attn_set_mode(0)
