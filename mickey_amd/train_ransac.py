"""Training-time vectorised RANSAC on the GPU: a drop-in for the reference's `MetricPoseLoss`
(lib/models/MicKey/modules/loss/loss_class.py:9-560) with the same constructor, attributes, method names and return
values.

Where the work goes:
  * outer sampling of NUM_SAMPLES_MATCHES cells per (pair, REINFORCE iteration) from final_scores (:136-137): the
    inference sampler, `mk_exprace_topk` (exponential race == torch.multinomial without replacement);
  * NUM_CORR_3d3d-point hypotheses and the <= NUM_REF_STEPS refinement rounds of all B*IT_MATCHES*IT_RANSAC hypotheses
    (:148-184, the no-grad block): `mk_train_ransac_masks`, one wave per hypothesis;
  * the REINFORCE scatter the reference runs as a python loop over B*IT_MATCHES rows (:251-261): `mk_reinforce_scatter`;
  * the part autograd must see -- back-projection of the sampled keypoints, the final masked Procrustes, soft inlier
    score, VCRE / pose loss, softmax aggregation (:139-146, :187-246) -- stays in torch on the device, written with the
    reference's operations so that `avg_loss.backward()` fills outputs['kps0'|'kps1'|'depth0'|'depth1'].grad as the
    reference's trainer expects (lib/models/MicKey/model.py:101-128).

There is no CPU path: tensors must live on the GPU and the HIP library must be loadable.
"""
import numpy as np
import torch

from . import ops
from ._native import MickeyHipError


def backproject_3d(uv, depth, K):
    """reference utils/training_utils.py:7-22 (differentiable w.r.t. uv and depth)."""
    ones = torch.ones((uv.shape[0], uv.shape[1], 1), device=uv.device, dtype=uv.dtype)
    return depth * (torch.linalg.inv(K) @ torch.cat([uv, ones], -1).transpose(2, 1)).transpose(2, 1)


def project_2d(P, K):
    """reference utils/training_utils.py:24-35."""
    q = (K @ P.transpose(2, 1)).transpose(2, 1)
    return (q / (q[:, :, 2:3] + 1e-16))[:, :, :2]


def weighted_procrustes_masked(A, Bp, w, eps=1e-16):
    """reference loss/solvers.py:13-26,45-52 with use_weights=True, use_mask=True: centroids weighted by w / (sum|w| + eps),
    covariance by the raw mask, R = V diag(1, 1, det(U V^T)) U^T, t = b_mean - a_mean R^T.  Differentiable."""
    wn = (w / (w.abs().sum(1, keepdim=True) + eps)).unsqueeze(-1)
    a_mean = (wn * A).sum(1, keepdim=True)
    b_mean = (wn * Bp).sum(1, keepdim=True)
    H = (A - a_mean).transpose(1, 2) @ (w.unsqueeze(-1) * (Bp - b_mean))
    U, _, V = torch.svd(H)
    Z = torch.eye(3, device=A.device, dtype=A.dtype).repeat(A.shape[0], 1, 1)
    Z[:, 2, 2] = torch.sign(torch.linalg.det(U @ V.transpose(1, 2)))
    R = V @ Z @ U.transpose(1, 2)
    return R, b_mean - a_mean @ R.transpose(1, 2), H


def soft_inlier_counting_3d(X0, X1, R, t, th):
    """reference utils/training_utils.py:55-61."""
    d = (((((R @ X0.transpose(2, 1)).transpose(2, 1) + t) - X1) ** 2.0).sum(-1) + 1e-6) ** 0.5
    return torch.sigmoid((5.0 / th) * (th - d)).sum(-1).view(X0.shape[0], 1)


def rot_angle_loss(R, Rgt):
    """reference loss/loss_utils.py:105-121."""
    tr = torch.diagonal(R.transpose(1, 2) @ Rgt, dim1=-2, dim2=-1).sum(-1)
    return torch.acos(torch.clip((tr - 1) / 2, -0.99999, 0.99999)).abs().unsqueeze(-1)


def trans_l1_loss(t, tgt):
    """reference loss/loss_utils.py:95-103."""
    return (t - tgt).abs().sum(-1)


_EYE = {}


def eye_grid(device):
    """The benchmark's 7 x 4 x 7 grid of virtual points (reference lib/benchmarks/reprojection.py:34-58), [196, 3]."""
    key = str(device)
    if key not in _EYE:
        x = (np.arange(7) - 3.0) * 0.3
        y = (np.arange(4) - 1.5) * 0.3
        z = np.arange(7).astype(float) * 0.3 + 1.8
        xx, yy, zz = np.meshgrid(x, y, z)
        _EYE[key] = torch.from_numpy(np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)).float().to(device)
    return _EYE[key]


def vcre_loss(R, t, Rgt, tgt, K, H=720):
    """reference lib/utils/metrics.py:56-80."""
    M = R.shape[0]
    E = eye_grid(R.device).unsqueeze(0).expand(M, -1, -1)
    uv_gt = project_2d(E, K)
    moved = R @ E.transpose(2, 1) + t.transpose(2, 1)
    resid = (Rgt.transpose(2, 1) @ moved - Rgt.transpose(2, 1) @ tgt.transpose(2, 1)).transpose(2, 1)
    uv = project_2d(resid, K)
    uv_gt, uv = torch.clip(uv_gt, 0, H), torch.clip(uv, 0, H)
    return ((((uv_gt - uv) ** 2.0).sum(-1) + 1e-6) ** 0.5).mean(-1).view(M, 1)


def compute_pose_loss(R, t, Rgt, tgt, K0=None, K1=None, soft_clipping=True):
    """reference loss/loss_utils.py:27-39."""
    l_rot, l_tr = rot_angle_loss(R, Rgt), trans_l1_loss(t, tgt)
    loss = torch.tanh(l_rot / 0.9) + torch.tanh(l_tr / 0.9) if soft_clipping else l_rot + l_tr
    return loss, l_rot, l_tr


def compute_vcre_loss(R, t, Rgt, tgt, K0, K1, soft_clipping=True):
    """reference loss/loss_utils.py:41-69: VCRE of the pose under K0 and of the inverse pose under K1, averaged."""
    Ri = R.transpose(2, 1)
    ti = (-1 * Ri @ t.transpose(2, 1)).transpose(2, 1)
    Rgi = Rgt.transpose(2, 1)
    tgi = (-1 * Rgi @ tgt.transpose(2, 1)).transpose(2, 1)
    loss = (vcre_loss(Ri, ti, Rgi, tgi, K1) + vcre_loss(R, t, Rgt, tgt, K0)) / 2.0
    if soft_clipping:
        loss = torch.tanh(loss / 80)
    return loss, rot_angle_loss(R, Rgt), trans_l1_loss(t, tgt)


class MetricPoseLoss(torch.nn.Module):
    """Same contract as the reference class (loss_class.py:9-70 for the configuration, :79-333 for the vectorised path).
    `forward(batch)` -> (avg_loss, outputs, [gradients [B, n, n]], num_valid_h)."""

    def __init__(self, cfg, seed=0):
        super().__init__()
        L = cfg.LOSS_CLASS if hasattr(cfg, "LOSS_CLASS") else cfg["LOSS_CLASS"]
        self.loss_type = L["LOSS_FUNCTION"]
        self.soft_clipping = L["SOFT_CLIPPING"]
        if self.loss_type == "POSE_ERR":
            self.compute_loss = compute_pose_loss
            sub = L["POSE_ERR"]
        elif self.loss_type == "VCRE":
            self.compute_loss = compute_vcre_loss
            sub = L["VCRE"]
        else:
            raise ValueError("LOSS_CLASS.LOSS_FUNCTION must be 'VCRE' or 'POSE_ERR', got %r" % (self.loss_type,))
        self.max_loss_null = sub["MAX_LOSS_SOFTVALUE"] if self.soft_clipping else sub["MAX_LOSS_VALUE"]
        self.num_samples_matches = int(L["SAMPLER"]["NUM_SAMPLES_MATCHES"])
        self.use_RANSAC_vectorized = True
        G = L["GENERATE_HYPOTHESES"]
        self.score_temperature = G["SCORE_TEMPERATURE"]
        self.it_matches = int(G["IT_MATCHES"])
        self.it_RANSAC = int(G["IT_RANSAC"])
        self.inlier_3d_th = G["INLIER_3D_TH"]
        self.inlier_ref_th = G["INLIER_REF_TH"]
        self.num_ref_steps = int(G["NUM_REF_STEPS"])
        self.num_corr_3d_3d = int(G["NUM_CORR_3d3d"])
        self.add_null_hypothesis = L["NULL_HYPOTHESIS"]["ADD_NULL_HYPOTHESIS"]
        self.th_outliers = L["NULL_HYPOTHESIS"]["TH_OUTLIERS"]
        C = L["CURRICULUM_LEARNING"]
        self.train_w_top = C["TRAIN_WITH_TOPK"] or C["TRAIN_CURRICULUM"]
        if C["TRAIN_CURRICULUM"]:
            self.topK = C["TOPK_INIT"]
        elif C["TRAIN_WITH_TOPK"]:
            self.topK = C["TOPK"]
        # Philox streams of the two samplers: (seed, 2 * call) and (seed, 2 * call + 1), keyed further by the pair index.
        # Under DDP every rank constructs this class with the same seed and advances _calls in lockstep, so without a
        # per-rank offset all ranks would draw IDENTICAL noise for their (different) pairs -- unbiased, but the REINFORCE
        # variance reduction of data parallelism would be lost.  When batch['pair_base'] is absent the pair index is
        # therefore offset by rank * 2^20 (read lazily: the process group usually does not exist yet here); a caller that
        # shards one global batch itself passes pair_base and gets sharding-invariant draws instead.
        self.seed = int(seed)
        self._calls = 0

    @staticmethod
    def _default_pair_base():
        import torch.distributed as dist
        return (dist.get_rank() << 20) if dist.is_available() and dist.is_initialized() else 0

    def read_pose_parameters(self, batch):
        """reference loss_class.py:71-78."""
        Rgt = batch["T_0to1"][:, :3, :3]
        tgt = batch["T_0to1"][:, :3, 3:].transpose(1, 2)
        return Rgt, tgt, batch["K_color0"].float(), batch["K_color1"].float()

    def single_iteration_RANSAC(self, batch, check_rank, idx_outer=None, idx_inner=None, return_debug=False):
        """reference loss_class.py:79-285.  idx_outer int [B*IT_MATCHES, S] / idx_inner int [B*IT_MATCHES*IT_RANSAC, NUM_CORR]
        replace the two draws (tests replay the reference's own torch.multinomial draws through them)."""
        matches = batch["final_scores"].detach()
        if not matches.is_cuda:
            raise MickeyHipError("MetricPoseLoss (mickey_amd) runs on the GPU only: batch['final_scores'] is on %s" % matches.device)
        kps0, depth0 = batch["kps0"].detach().requires_grad_(), batch["depth_kp0"].detach().requires_grad_()
        kps1, depth1 = batch["kps1"].detach().requires_grad_(), batch["depth_kp1"].detach().requires_grad_()
        dev = matches.device
        B, n, _ = matches.shape
        ncell = n * n
        it_m, it_r, S, nc = self.it_matches, self.it_RANSAC, self.num_samples_matches, self.num_corr_3d_3d
        Ro, Ri = B * it_m, B * it_m * it_r
        rowp = matches.reshape(B, ncell).float()
        outputs = {"kps0": kps0, "kps1": kps1, "depth0": depth0, "depth1": depth1}
        baseline = torch.zeros((B,), device=dev)
        losses_rot = torch.zeros((B, 1), device=dev)
        losses_trans = torch.zeros((B, 1), device=dev)
        gradients, gradients_b = torch.zeros_like(rowp), torch.zeros_like(rowp)
        dbg = {}

        def bail():
            out = (baseline, losses_rot, losses_trans, gradients, gradients_b, outputs, 0)
            return out + (dbg,) if return_debug else out

        call = self._calls
        self._calls += 1
        if idx_outer is None:
            # the sampler's own input scan raises `invalid` on NaN / inf / negative cells and on empty rows: the cases in
            # which the reference skips the loop (:118-124) or lands in its except branch (:263-270)
            invalid = torch.zeros((1,), device=dev, dtype=torch.int32)
            idx_outer, cnt = ops.exprace_topk(rowp, it_m, S, seed=self.seed, offset=2 * call, invalid=invalid,
                                              pair_base=int(batch.get("pair_base", self._default_pair_base())))
            if int(invalid.item()) != 0 or int((cnt < S).any().item()) != 0:
                print("Invalid matching matrix! Skip RANSAC loop.")
                return bail()
        idx_outer = idx_outer.to(device=dev, dtype=torch.int64)
        pair_of_row = torch.arange(B, device=dev).repeat_interleave(it_m)
        bo = pair_of_row.view(Ro, 1).expand(Ro, S)
        i0, i1 = torch.div(idx_outer, n, rounding_mode="trunc"), idx_outer % n
        cor0, cor1 = kps0[bo, :2, i0], kps1[bo, :2, i1]
        d0, d1 = depth0[bo, :2, i0], depth1[bo, :2, i1]
        weights = rowp[bo, idx_outer]
        Rgt, tgt, K0, K1 = self.read_pose_parameters(batch)
        X = backproject_3d(cor0, d0, K0[pair_of_row])
        Y = backproject_3d(cor1, d1, K1[pair_of_row])
        # hypotheses + refinement, no autograd (the reference wraps the same steps in torch.no_grad, :152-184)
        mask, idx_in, rounds = ops.train_ransac_masks(
            X.detach(), Y.detach(), weights, it_r, float(self.inlier_ref_th), self.num_ref_steps, nc, idx_in=idx_inner,
            seed=self.seed, offset=2 * call + 1, set_base=int(batch.get("pair_base", self._default_pair_base())) * it_m)
        X_v = X.unsqueeze(1).expand(Ro, it_r, S, 3).reshape(Ri, S, 3)
        Y_v = Y.unsqueeze(1).expand(Ro, it_r, S, 3).reshape(Ri, S, 3)
        R, t, H = weighted_procrustes_masked(X_v, Y_v, mask)
        dbg.update(idx_outer=idx_outer, idx_inner=idx_in, inliers_final=mask, rounds=rounds, R=R.detach(), t=t.detach())
        if check_rank and int((torch.linalg.matrix_rank(H.detach()) == 1).sum().item()) > 0:
            print("[ERROR]: Skipping RANSAC iteration due to rank matrix.")
            return bail()
        if not bool(torch.isfinite(R).all().item() and torch.isfinite(t).all().item()):
            print("[ERROR]: Skipping RANSAC iteration due to invalid values in R/t.")
            return bail()
        score_k = soft_inlier_counting_3d(X_v, Y_v, R, t, th=self.inlier_3d_th)
        pair_of_hyp = torch.arange(B, device=dev).repeat_interleave(it_m * it_r)
        loss_value_k, loss_rot_k, loss_trans_k = self.compute_loss(
            R, t, Rgt.float()[pair_of_hyp], tgt.float()[pair_of_hyp], batch["Kori_color0"].float()[pair_of_hyp],
            batch["Kori_color1"].float()[pair_of_hyp], soft_clipping=self.soft_clipping)
        loss_value_k, loss_rot_k, loss_trans_k, score_k = (v.reshape(Ro, it_r) for v in (loss_value_k, loss_rot_k, loss_trans_k, score_k))
        sm = torch.softmax(score_k / self.score_temperature, -1)
        loss_rot = (loss_rot_k * sm).sum(-1).unsqueeze(-1)
        loss_trans = (loss_trans_k * sm).sum(-1).unsqueeze(-1)
        if self.add_null_hypothesis:
            loss_value_k = torch.cat([loss_value_k, torch.full((Ro, 1), float(self.max_loss_null), device=dev)], -1)
            score_k = torch.cat([score_k, torch.full((Ro, 1), float(self.th_outliers * S), device=dev)], -1)
        loss_value = (loss_value_k * torch.softmax(score_k / self.score_temperature, -1)).sum(-1).unsqueeze(-1)
        gradients, gradients_b = ops.reinforce_scatter(idx_outer, loss_value.detach().reshape(Ro), B, it_m, ncell)
        losses_rot = loss_rot.reshape(B, it_m).sum(-1).unsqueeze(-1)
        losses_trans = loss_trans.reshape(B, it_m).sum(-1).unsqueeze(-1)
        baseline = loss_value.reshape(B, it_m).sum(-1)
        dbg.update(loss_value=loss_value.detach(), score=score_k.detach())
        out = (baseline, losses_rot, losses_trans, gradients, gradients_b, outputs, 1)
        return out + (dbg,) if return_debug else out

    def RANSAC_vectorized(self, batch, check_rank=False, idx_outer=None, idx_inner=None):
        """reference loss_class.py:287-333."""
        B, n, _ = batch["final_scores"].shape
        baseline, losses_rot, losses_trans, gradients, gradients_b, outputs, num_valid_h = \
            self.single_iteration_RANSAC(batch, check_rank, idx_outer, idx_inner)
        baseline = baseline / self.it_matches
        losses_trans = losses_trans / self.it_matches
        losses_rot = losses_rot / self.it_matches
        gradients = (gradients - gradients_b * baseline.detach().view(B, 1)) / self.it_matches
        if num_valid_h == 0:
            print("[ERROR]: No valid hypotheses generated!")
        if self.train_w_top and B > 1:
            select_topB = max(int(B * self.topK / 100), 1)
            topk_loss = baseline[torch.argsort(baseline)[select_topB]]
            mask_topk = (baseline < topk_loss).float()
            avg_loss = (mask_topk * baseline).sum() / mask_topk.sum()
            gradients = gradients * mask_topk.unsqueeze(-1)
        else:
            avg_loss = torch.mean(baseline)
            mask_topk = torch.ones(B, device=baseline.device)
        outputs["avg_loss_rot"] = torch.mean(losses_rot)
        outputs["avg_loss_trans"] = torch.mean(losses_trans)
        outputs["avg_rot_errs"] = torch.mean(torch.rad2deg(losses_rot.detach()))
        outputs["avg_t_errs"] = torch.mean(losses_trans)
        outputs["mask_topk"] = mask_topk
        return avg_loss, outputs, [gradients.reshape(B, n, n)], num_valid_h

    def forward(self, batch):
        return self.RANSAC_vectorized(batch)
