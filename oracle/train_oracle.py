"""CPU restatement of the reference's training-time vectorised RANSAC.

TEST INFRASTRUCTURE -- imported only by tests/, tools/ benchmarks' CPU leg and oracle/make_golden_train.py; the product
(mickey_amd/train_ransac.py) never touches it.

What it follows (reference = /root/reference):
  lib/models/MicKey/modules/loss/loss_class.py:79-285   MetricPoseLoss.single_iteration_RANSAC
  lib/models/MicKey/modules/loss/loss_class.py:287-333  MetricPoseLoss.RANSAC_vectorized
  lib/models/MicKey/modules/loss/loss_utils.py:27-69    compute_pose_loss / compute_vcre_loss
  lib/models/MicKey/modules/loss/loss_utils.py:95-121   trans_l1_loss / rot_angle_loss
  lib/utils/metrics.py:56-80                            vcre_loss
  lib/benchmarks/reprojection.py:34-58                  get_grid_multipleheight (the 7 x 4 x 7 virtual-point grid)

Parity pin: oracle/make_golden_train.py imports the reference's MetricPoseLoss in this container with torch.multinomial
replaced by a replay of recorded draws and stores inputs, draws and every output (incl. the keypoint / depth gradients of
avg_loss.backward()) in tests/golden/train_ransac.npz; tests/test_train_oracle.py checks this file against them.

The per-hypothesis refinement is written as the state machine the reference's masked tensor updates amount to:
    cur = the 8 sampled matches, final = cur, pre = 8, active = True
    repeat NUM_REF_STEPS times, for active hypotheses only:
        (R, t) = masked Procrustes over cur;  ref = {matches within INLIER_REF_TH of (R, t)}
        active = |ref| > pre;  if active: pre = |ref|, final = cur, cur = ref
so `final` is the match set that PRODUCED the last accepted pose (loss_class.py:169-181), and the differentiable pose is
the masked Procrustes over `final` (:187).
"""
import math

import numpy as np
import torch

from . import mickey_oracle as mo


def eye_grid():
    """[196, 3] virtual points in front of the camera: x in 7 steps of 0.3 centred, y in 4 steps of 0.3 centred, z from 1.8
    in 7 steps of 0.3, meshgrid order (y-major, then x, then z).  reference lib/benchmarks/reprojection.py:34-58."""
    x = (np.arange(7) - 3.0) * 0.3
    y = (np.arange(4) - 1.5) * 0.3
    z = np.arange(7).astype(float) * 0.3 + 1.8
    xx, yy, zz = np.meshgrid(x, y, z)
    return torch.from_numpy(np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)).float()


def project_2d(P, K):
    """reference utils/training_utils.py:24-35."""
    q = (K @ P.transpose(2, 1)).transpose(2, 1)
    return (q / (q[:, :, 2:3] + 1e-16))[:, :, :2]


def vcre(R, t, Rgt, tgt, K, H=720):
    """Mean reprojection distance of the virtual grid between the estimated and the ground-truth pose, both clipped to
    [0, H].  reference lib/utils/metrics.py:56-80.  R, Rgt [M,3,3]; t, tgt [M,1,3]."""
    M = R.shape[0]
    E = eye_grid().unsqueeze(0).repeat(M, 1, 1)
    uv_gt = project_2d(E, K)
    moved = R @ E.transpose(2, 1) + t.transpose(2, 1)
    resid = (Rgt.transpose(2, 1) @ moved - Rgt.transpose(2, 1) @ tgt.transpose(2, 1)).transpose(2, 1)
    uv = project_2d(resid, K)
    uv_gt, uv = torch.clip(uv_gt, 0, H), torch.clip(uv, 0, H)
    return (((uv_gt - uv) ** 2.0).sum(-1) + 1e-6).pow(0.5).mean(-1).view(M, 1)


def rot_angle_loss(R, Rgt):
    """reference loss_utils.py:105-121."""
    tr = torch.diagonal(R.transpose(1, 2) @ Rgt, dim1=-2, dim2=-1).sum(-1)
    return torch.acos(torch.clip((tr - 1) / 2, -0.99999, 0.99999)).abs().unsqueeze(-1)


def trans_l1_loss(t, tgt):
    """reference loss_utils.py:95-103."""
    return (t - tgt).abs().sum(-1)


def pose_loss(kind, R, t, Rgt, tgt, K0, K1, soft_clipping):
    """kind 'VCRE': symmetric VCRE (pose and inverse pose), tanh(x/80) when soft-clipped (loss_utils.py:41-69);
    kind 'POSE_ERR': rotation angle + L1 translation, each tanh(x/0.9) when soft-clipped (:27-39)."""
    l_rot, l_tr = rot_angle_loss(R, Rgt), trans_l1_loss(t, tgt)
    if kind == "POSE_ERR":
        loss = torch.tanh(l_rot / 0.9) + torch.tanh(l_tr / 0.9) if soft_clipping else l_rot + l_tr
        return loss, l_rot, l_tr
    Ri = R.transpose(2, 1)
    ti = (-1 * Ri @ t.transpose(2, 1)).transpose(2, 1)
    Rgi = Rgt.transpose(2, 1)
    tgi = (-1 * Rgi @ tgt.transpose(2, 1)).transpose(2, 1)
    loss = (vcre(Ri, ti, Rgi, tgi, K1) + vcre(R, t, Rgt, tgt, K0)) / 2.0
    if soft_clipping:
        loss = torch.tanh(loss / 80)
    return loss, l_rot, l_tr


def loss_constants(cfg):
    L = cfg["LOSS_CLASS"]
    G = L["GENERATE_HYPOTHESES"]
    kind = L["LOSS_FUNCTION"]
    soft = bool(L["SOFT_CLIPPING"])
    sub = L["POSE_ERR"] if kind == "POSE_ERR" else L["VCRE"]
    cur = L["CURRICULUM_LEARNING"]
    top = None
    if cur["TRAIN_CURRICULUM"]:
        top = cur["TOPK_INIT"]
    elif cur["TRAIN_WITH_TOPK"]:
        top = cur["TOPK"]
    return dict(kind=kind, soft=soft, max_null=sub["MAX_LOSS_SOFTVALUE"] if soft else sub["MAX_LOSS_VALUE"],
                S=int(L["SAMPLER"]["NUM_SAMPLES_MATCHES"]), temp=float(G["SCORE_TEMPERATURE"]), it_m=int(G["IT_MATCHES"]),
                it_r=int(G["IT_RANSAC"]), th3d=float(G["INLIER_3D_TH"]), thref=float(G["INLIER_REF_TH"]),
                nref=int(G["NUM_REF_STEPS"]), nc=int(G["NUM_CORR_3d3d"]), null=bool(L["NULL_HYPOTHESIS"]["ADD_NULL_HYPOTHESIS"]),
                th_out=float(L["NULL_HYPOTHESIS"]["TH_OUTLIERS"]),
                train_w_top=bool(cur["TRAIN_WITH_TOPK"] or cur["TRAIN_CURRICULUM"]), topK=top)


def refine_masks(Xv, Yv, idx_inner, nref, thref, nc):
    """The no-grad part of loss_class.py:152-184.  Xv, Yv [M,S,3]; idx_inner [M,nc] -> (final mask [M,S] float, rounds [M])."""
    M, S, _ = Xv.shape
    rows = torch.arange(M).view(M, 1)
    cur = torch.zeros(M, S)
    cur[rows, idx_inner] = 1
    final = cur.clone()
    pre = torch.full((M,), float(nc))
    active = torch.ones(M, dtype=torch.bool)
    rounds = torch.zeros(M, dtype=torch.int32)
    for _ in range(nref):
        if int(active.sum()) == 0:
            break
        R, t, _ = mo.kabsch(Xv[active], Yv[active], cur[active], masked=True)
        ref = mo.hard_inliers(Xv[active], Yv[active], R, t, thref)
        cnt = ref.sum(-1)
        ok = cnt > pre[active]
        ia = torch.nonzero(active).view(-1)
        go = ia[ok]
        pre[go] = cnt[ok]
        final[go] = cur[go]
        cur[go] = ref[ok]
        rounds[go] += 1
        active = torch.zeros(M, dtype=torch.bool)
        active[go] = True
    return final, rounds


def single_iteration(batch, cfg, idx_outer=None, idx_inner=None, generator=None):
    """MetricPoseLoss.single_iteration_RANSAC (loss_class.py:79-285).  idx_outer [B*it_m, S] / idx_inner [B*it_m*it_r, nc]
    replace the two torch.multinomial draws when given.  Returns the reference's 7-tuple plus a debug dict."""
    c = loss_constants(cfg)
    matches = batch["final_scores"].detach().float()
    kps0, dep0 = batch["kps0"].detach().clone().requires_grad_(), batch["depth_kp0"].detach().clone().requires_grad_()
    kps1, dep1 = batch["kps1"].detach().clone().requires_grad_(), batch["depth_kp1"].detach().clone().requires_grad_()
    B, n, _ = matches.shape
    ncell = n * n
    it_m, it_r, S, nc = c["it_m"], c["it_r"], c["S"], c["nc"]
    Ro, Ri = B * it_m, B * it_m * it_r
    rowp = matches.reshape(B, ncell)
    outputs = {"kps0": kps0, "kps1": kps1, "depth0": dep0, "depth1": dep1}
    baseline = torch.zeros(B)
    losses_rot, losses_trans = torch.zeros(B, 1), torch.zeros(B, 1)
    grads, grads_b = torch.zeros_like(rowp), torch.zeros_like(rowp)
    dbg = {}
    bad = bool(torch.isnan(rowp).any() or torch.isinf(rowp).any() or (rowp < 0).any())
    if bad:
        return baseline, losses_rot, losses_trans, grads, grads_b, outputs, 0, dbg
    pair_of_row = torch.arange(B).repeat_interleave(it_m)
    if idx_outer is None:
        try:
            idx_outer = torch.multinomial(rowp[pair_of_row], S, generator=generator)
        except RuntimeError:
            return baseline, losses_rot, losses_trans, grads, grads_b, outputs, 0, dbg
    idx_outer = idx_outer.long()
    i0, i1 = torch.div(idx_outer, n, rounding_mode="trunc"), idx_outer % n
    bo = pair_of_row.view(Ro, 1).expand(Ro, S)
    cor0, cor1 = kps0[bo, :2, i0], kps1[bo, :2, i1]
    d0, d1 = dep0[bo, :2, i0], dep1[bo, :2, i1]
    w = rowp[bo, idx_outer]
    K0, K1 = batch["K_color0"].float(), batch["K_color1"].float()
    X = mo.backproject(cor0, d0, K0[pair_of_row])
    Y = mo.backproject(cor1, d1, K1[pair_of_row])
    Xv = X.unsqueeze(1).expand(Ro, it_r, S, 3).reshape(Ri, S, 3)
    Yv = Y.unsqueeze(1).expand(Ro, it_r, S, 3).reshape(Ri, S, 3)
    if idx_inner is None:
        wv = w.unsqueeze(1).expand(Ro, it_r, S).reshape(Ri, S)
        try:
            idx_inner = torch.multinomial(wv, nc, generator=generator)
        except RuntimeError:
            return baseline, losses_rot, losses_trans, grads, grads_b, outputs, 0, dbg
    idx_inner = idx_inner.long()
    with torch.no_grad():
        final, rounds = refine_masks(Xv.detach(), Yv.detach(), idx_inner, c["nref"], c["thref"], nc)
    R, t, _ = mo.kabsch(Xv, Yv, final, masked=True)
    dbg.update(idx_outer=idx_outer, idx_inner=idx_inner, inliers_final=final, rounds=rounds, R=R.detach(), t=t.detach())
    if not bool(torch.isfinite(R).all() and torch.isfinite(t).all()):
        return baseline, losses_rot, losses_trans, grads, grads_b, outputs, 0, dbg
    score = mo.soft_inliers(Xv, Yv, R, t, c["th3d"])
    pair_of_hyp = torch.arange(B).repeat_interleave(it_m * it_r)
    T = batch["T_0to1"].float()
    Rgt, tgt = T[:, :3, :3][pair_of_hyp], T[:, :3, 3:].transpose(1, 2)[pair_of_hyp]
    lv, lr, lt = pose_loss(c["kind"], R, t, Rgt, tgt, batch["Kori_color0"].float()[pair_of_hyp],
                           batch["Kori_color1"].float()[pair_of_hyp], c["soft"])
    lv, lr, lt, score = (v.reshape(Ro, it_r) for v in (lv, lr, lt, score))
    sm = torch.softmax(score / c["temp"], -1)
    loss_rot = (lr * sm).sum(-1, keepdim=True)
    loss_trans = (lt * sm).sum(-1, keepdim=True)
    if c["null"]:
        lv = torch.cat([lv, torch.full((Ro, 1), float(c["max_null"]))], -1)
        score = torch.cat([score, torch.full((Ro, 1), c["th_out"] * S)], -1)
    loss_value = (lv * torch.softmax(score / c["temp"], -1)).sum(-1, keepdim=True)
    # REINFORCE bookkeeping, row after row as the reference's loop does (:251-261); fp32 sums in that order
    lvd = loss_value.detach()
    for r in range(Ro):
        b = int(pair_of_row[r])
        grads_b[b, idx_outer[r]] += 1
        grads[b, idx_outer[r]] += lvd[r, 0]
    losses_rot = loss_rot.reshape(B, it_m).sum(-1, keepdim=True)
    losses_trans = loss_trans.reshape(B, it_m).sum(-1, keepdim=True)
    baseline = loss_value.reshape(B, it_m).sum(-1)
    dbg.update(loss_value=lvd, score=score.detach())
    return baseline, losses_rot, losses_trans, grads, grads_b, outputs, 1, dbg


def ransac_vectorized(batch, cfg, idx_outer=None, idx_inner=None, generator=None):
    """MetricPoseLoss.RANSAC_vectorized (loss_class.py:287-333): expected loss, baseline-subtracted gradients, curriculum
    top-K mask.  Returns (avg_loss, outputs, [gradients [B,n,n]], num_valid_h, debug)."""
    c = loss_constants(cfg)
    B, n, _ = batch["final_scores"].shape
    baseline, l_rot, l_tr, grads, grads_b, outputs, nvalid, dbg = single_iteration(batch, cfg, idx_outer, idx_inner, generator)
    baseline, l_tr, l_rot = baseline / c["it_m"], l_tr / c["it_m"], l_rot / c["it_m"]
    grads = (grads - grads_b * baseline.view(B, 1)) / c["it_m"]
    if c["train_w_top"] and B > 1:
        sel = max(int(B * c["topK"] / 100), 1)
        cut = baseline[torch.argsort(baseline)[sel]]
        mask = (baseline < cut).float()
        avg = (mask * baseline).sum() / mask.sum()
        grads = grads * mask.unsqueeze(-1)
    else:
        avg = baseline.mean()
        mask = torch.ones(B)
    outputs.update(avg_loss_rot=l_rot.mean(), avg_loss_trans=l_tr.mean(), avg_rot_errs=torch.rad2deg(l_rot.detach()).mean(),
                   avg_t_errs=l_tr.mean(), mask_topk=mask)
    return avg, outputs, [grads.reshape(B, n, n)], nvalid, dbg


def default_loss_cfg():
    """LOSS_CLASS of the reference's config/MicKey/curriculum_learning.yaml:55-87."""
    return {"LOSS_CLASS": {
        "LOSS_FUNCTION": "VCRE", "SOFT_CLIPPING": True,
        "POSE_ERR": {"MAX_LOSS_VALUE": 1.5, "MAX_LOSS_SOFTVALUE": 0.8},
        "VCRE": {"MAX_LOSS_VALUE": 90, "MAX_LOSS_SOFTVALUE": 0.8},
        "GENERATE_HYPOTHESES": {"SCORE_TEMPERATURE": 20, "IT_MATCHES": 20, "IT_RANSAC": 20, "INLIER_3D_TH": 0.3,
                                "INLIER_REF_TH": 0.15, "NUM_REF_STEPS": 4, "NUM_CORR_3d3d": 8},
        "NULL_HYPOTHESIS": {"ADD_NULL_HYPOTHESIS": True, "TH_OUTLIERS": 0.35},
        "CURRICULUM_LEARNING": {"TRAIN_CURRICULUM": True, "TRAIN_WITH_TOPK": True, "TOPK_INIT": 30, "TOPK": 80},
        "SAMPLER": {"NUM_SAMPLES_MATCHES": 512}}}


def synthetic_batch(B, n, seed, inlier_frac=0.5, noise=0.03):
    """A training batch with a planted relative pose: keypoints / depths of image 0 are random, a fraction of image 1's
    are their projections under the ground-truth pose (so hypotheses refine), final_scores favour the true matches."""
    g = torch.Generator().manual_seed(seed)
    K = torch.tensor([[590.0, 0, 270.0], [0, 590.0, 360.0], [0, 0, 1.0]]).repeat(B, 1, 1)
    kps0 = torch.rand(B, 2, n, generator=g) * torch.tensor([540.0, 720.0]).view(1, 2, 1)
    dep0 = 1.0 + 4.0 * torch.rand(B, 1, n, generator=g)
    ang = 0.3 * (torch.rand(B, 3, generator=g) - 0.5)
    Rgt = torch.stack([_rodrigues(a) for a in ang])
    tgt = 0.6 * (torch.rand(B, 1, 3, generator=g) - 0.5)
    X0 = mo.backproject(kps0.transpose(1, 2), dep0.transpose(1, 2), K)
    X1 = (Rgt @ X0.transpose(2, 1)).transpose(2, 1) + tgt
    uv1 = project_2d(X1, K)
    kps1 = uv1.transpose(1, 2).clone()
    dep1 = X1[:, :, 2:3].transpose(1, 2).clone()
    nin = int(n * inlier_frac)
    kps1[:, :, nin:] = torch.rand(B, 2, n - nin, generator=g) * torch.tensor([540.0, 720.0]).view(1, 2, 1)
    dep1[:, :, nin:] = 1.0 + 4.0 * torch.rand(B, 1, n - nin, generator=g)
    dep1 = dep1 + noise * torch.randn(B, 1, n, generator=g)
    scores = 1e-4 * torch.rand(B, n, n, generator=g)
    diag = torch.arange(nin)
    scores[:, diag, diag] += 0.2 + 0.6 * torch.rand(B, nin, generator=g)
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, :3, :3] = Rgt
    T[:, :3, 3] = tgt[:, 0]
    return {"final_scores": scores, "kps0": kps0, "kps1": kps1, "depth_kp0": dep0, "depth_kp1": dep1, "T_0to1": T,
            "K_color0": K.clone(), "K_color1": K.clone(), "Kori_color0": K.clone(), "Kori_color1": K.clone()}


def _rodrigues(a):
    th = float(a.norm())
    if th < 1e-12:
        return torch.eye(3)
    k = a / th
    Kx = torch.tensor([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return torch.eye(3) + math.sin(th) * Kx + (1 - math.cos(th)) * (Kx @ Kx)
