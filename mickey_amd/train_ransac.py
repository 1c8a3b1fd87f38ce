"""Training-time vectorised RANSAC on the GPU: a drop-in for the reference's `MetricPoseLoss`
(lib/models/MicKey/modules/loss/loss_class.py:9-560) with the same constructor, attributes, method names and return
values.

Where the work goes:
  * outer sampling of NUM_SAMPLES_MATCHES cells per (pair, REINFORCE iteration) from final_scores (:136-137): the
    inference sampler, `mk_exprace_topk` (exponential race == torch.multinomial without replacement);
  * NUM_CORR_3d3d-point hypotheses and the <= NUM_REF_STEPS refinement rounds of all B*IT_MATCHES*IT_RANSAC hypotheses
    (:148-184, the no-grad block): `mk_train_ransac_masks`, one wave per hypothesis;
  * the REINFORCE scatter the reference runs as a python loop over B*IT_MATCHES rows (:251-261): `mk_reinforce_scatter`;
  * the differentiable tail -- the final masked Procrustes of every hypothesis, its soft inlier score and its VCRE / pose
    loss (:187-246) -- is a torch.autograd.Function whose forward AND backward are HIP kernels (`mk_train_tail_fwd` /
    `mk_train_tail_bwd`: one wave per hypothesis, the 3x3 SVD differentiated in closed form);
  * what is left to torch: the gather + back-projection of the sampled keypoints (:139-146) and the softmax aggregation over
    the hypotheses of a set (tiny [B*IT_MATCHES, IT_RANSAC] tensors), so that `avg_loss.backward()` fills
    outputs['kps0'|'kps1'|'depth0'|'depth1'].grad as the reference's trainer expects (lib/models/MicKey/model.py:101-128).

There is no CPU path: tensors must live on the GPU and the HIP library must be loadable.
"""
import torch

from . import ops
from ._native import MickeyHipError


def backproject_3d(uv, depth, K):
    """reference utils/training_utils.py:7-22 (differentiable w.r.t. uv and depth)."""
    ones = torch.ones((uv.shape[0], uv.shape[1], 1), device=uv.device, dtype=uv.dtype)
    return depth * (torch.linalg.inv(K) @ torch.cat([uv, ones], -1).transpose(2, 1)).transpose(2, 1)


class _RansacTail(torch.autograd.Function):
    """reference loss_class.py:187-246 -- the masked Procrustes of every hypothesis, its soft inlier score and its loss -- as
    two HIP entry points (mk_train_tail_fwd / mk_train_tail_bwd, csrc/mk_train_tail.hip): forward AND backward; the SVD is
    differentiated in closed form.  Inputs that the reference detaches (mask, ground truth, intrinsics) get no gradient."""

    @staticmethod
    def forward(ctx, X, Y, mask, Rgt, tgt, K0, K1, it_r, it_m, th, loss_type, soft):
        out, Rt, saved = ops.train_tail_fwd(X, Y, mask, Rgt, tgt, K0, K1, it_r, it_m, th, loss_type, soft)
        ctx.save_for_backward(X, Y, mask, Rgt, tgt, K0, K1, Rt, saved)
        ctx.cfg = (it_r, it_m, th, loss_type, soft)
        loss_value, loss_rot, loss_trans, score = (out[:, i].contiguous() for i in range(4))
        ctx.mark_non_differentiable(loss_rot, loss_trans, Rt, saved)
        return loss_value, loss_rot, loss_trans, score, Rt, saved

    @staticmethod
    def backward(ctx, g_loss, g_rot, g_trans, g_score, g_Rt, g_saved):
        X, Y, mask, Rgt, tgt, K0, K1, Rt, saved = ctx.saved_tensors
        it_r, it_m, th, loss_type, soft = ctx.cfg
        zero = torch.zeros((Rt.shape[0],), device=X.device)
        g = torch.stack([g_loss if g_loss is not None else zero, g_score if g_score is not None else zero], 1).float()
        gX, gY = ops.train_tail_bwd(X, Y, mask, Rgt, tgt, K0, K1, it_r, it_m, th, loss_type, soft, Rt, saved, g)
        return (gX, gY) + (None,) * 10


class MetricPoseLoss(torch.nn.Module):
    """Same contract as the reference class (loss_class.py:9-70 for the configuration, :79-333 for the vectorised path).
    `forward(batch)` -> (avg_loss, outputs, [gradients [B, n, n]], num_valid_h)."""

    def __init__(self, cfg, seed=0):
        super().__init__()
        L = cfg.LOSS_CLASS if hasattr(cfg, "LOSS_CLASS") else cfg["LOSS_CLASS"]
        self.loss_type = L["LOSS_FUNCTION"]
        self.soft_clipping = L["SOFT_CLIPPING"]
        if self.loss_type == "POSE_ERR":
            sub = L["POSE_ERR"]
        elif self.loss_type == "VCRE":
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
        # the differentiable tail (reference :187-246): masked Procrustes, soft inlier score and loss of every hypothesis,
        # forward and backward in HIP; autograd continues from dL/dX, dL/dY into the back-projection above
        pair_b = lambda v, w: v.float().reshape(B, w).contiguous()  # noqa: E731
        vcre = self.loss_type == "VCRE"
        loss_value_k, loss_rot_k, loss_trans_k, score_k, Rt, saved = _RansacTail.apply(
            X, Y, mask, pair_b(Rgt, 9), pair_b(tgt, 3), pair_b(batch["Kori_color0"], 9) if vcre else None,
            pair_b(batch["Kori_color1"], 9) if vcre else None, it_r, it_m, float(self.inlier_3d_th), 0 if vcre else 1,
            bool(self.soft_clipping))
        R, t = Rt[:, :9].reshape(Ri, 3, 3), Rt[:, 9:].reshape(Ri, 1, 3)
        dbg.update(idx_outer=idx_outer, idx_inner=idx_in, inliers_final=mask, rounds=rounds, R=R, t=t)
        if check_rank:
            # torch.linalg.matrix_rank(H) == 1 (reference :190): singular values above max(S) * 3 * eps(fp32)
            sv = saved[:, 18:21]
            if int(((sv > sv[:, :1] * 3 * 1.1920929e-07).sum(1) == 1).sum().item()) > 0:
                print("[ERROR]: Skipping RANSAC iteration due to rank matrix.")
                return bail()
        if not bool(torch.isfinite(Rt).all().item()):
            print("[ERROR]: Skipping RANSAC iteration due to invalid values in R/t.")
            return bail()
        loss_value_k, loss_rot_k, loss_trans_k, score_k = (v.reshape(Ro, it_r) for v in (loss_value_k, loss_rot_k, loss_trans_k, score_k))
        sm = torch.softmax(score_k / self.score_temperature, -1)
        # rotation / translation errors are LOGGING outputs here (the reference logs them, model.py:151-171, and optimises
        # avg_loss only): the native tail returns them without a gradient, so they are aggregated with a DETACHED softmax --
        # backpropagating avg_loss_rot / avg_loss_trans raises instead of silently yielding the partial gradient through `sm`
        loss_rot = (loss_rot_k * sm.detach()).sum(-1).unsqueeze(-1)
        loss_trans = (loss_trans_k * sm.detach()).sum(-1).unsqueeze(-1)
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
