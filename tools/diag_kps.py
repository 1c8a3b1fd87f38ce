"""Which kernel moves the bf16 keypoints (advisor item of round 3: test_full_forward_golden[bf16] measured a 0.12 px maximum
keypoint deviation where round 2 measured < 0.1)?  The golden 182x196 case under every switch that selects another kernel:
classic-softmax attention (attn mode 3), no row centring, no LayerNorm fold (stand-alone LayerNorm kernels + fp32 stream),
and the head stacks on bf16 / fp16 / fp32 operands.  Prints kps rel-Frobenius, the maximum deviation in pixels and dsc / scr
errors per variant, next to the oracle-side floors."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mickey_amd import ops, synthetic as syn  # noqa: E402
from mickey_amd.config import default_cfg  # noqa: E402
from mickey_amd.model import MickeyRelativePose  # noqa: E402


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "full_forward.npz")))
    fl = dict(np.load(os.path.join(ROOT, "tests", "golden", "noise_floor_lp.npz")))
    cfg = default_cfg()
    sd = syn.mickey_state_dict(cfg, seed=0)
    batch = syn.synthetic_batch(B=2, H=182, W=196, seed=1234)
    variants = [("bf16 heads bf16 (round-3 configuration)", dict(HEADS_DTYPE="same"), 0),
                ("  + classic-softmax attention (mode 3)", dict(HEADS_DTYPE="same"), 3),
                ("  + no row centring", dict(HEADS_DTYPE="same", LN_CENTRE=False), 0),
                ("  + no LayerNorm fold", dict(HEADS_DTYPE="same", LN_FOLD=False), 0),
                ("bf16 encoder, heads fp16 (auto)", dict(HEADS_DTYPE="auto"), 0),
                ("bf16 encoder, heads fp32", dict(HEADS_DTYPE="fp32"), 0)]
    print("oracle-side floors at 182x196: all-bf16 kps %.2e (max %.3f px) dsc %.2e scr %.2e | bf16 encoder only kps %.2e (max %.3f px) "
          "dsc %.2e scr %.2e" % (fl["bf16_encheads_182_kps0"], fl["bf16_encheads_182_kps0_maxabs"], fl["bf16_encheads_182_dsc0"],
                                 fl["bf16_encheads_182_scr0"], fl["bf16_enc_182_kps0"], fl["bf16_enc_182_kps0_maxabs"],
                                 fl["bf16_enc_182_dsc0"], fl["bf16_enc_182_scr0"]))
    for name, amd, mode in variants:
        c = copy.deepcopy(cfg)
        c["AMD"]["ENCODER_DTYPE"] = "bf16"
        c["AMD"].update(amd)
        m = MickeyRelativePose(c)
        m.load_state_dict(sd)
        m = m.cuda()
        ops.attn_set_mode(mode)
        d = {k: v.cuda() for k, v in batch.items()}
        m.compute_correspondences(d)
        ops.attn_set_mode(0)
        kmax = float((d["kps0"].cpu() - torch.from_numpy(g["kps0"])).abs().max())
        print("%-44s kps %.2e  max %.3f px  depth %.2e  scr %.2e  dsc %.2e  scores %.2e  final %.2e" % (
            name, rel(d["kps0"], g["kps0"]), kmax, rel(d["depth_kp0"], g["depth_kp0"]), rel(d["scr0"], g["scr0"]),
            rel(d["dsc0"], g["dsc0"]), rel(d["scores"], g["scores"]), rel(d["final_scores"], g["final_scores"])))
        del m, d
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
