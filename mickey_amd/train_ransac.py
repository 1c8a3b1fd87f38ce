"""Training-time vectorised RANSAC on the GPU: a drop-in for the reference's `MetricPoseLoss`
(lib/models/MicKey/modules/loss/loss_class.py:9-560) with the same constructor, attributes, method names and return
values.

Where the work goes:
  * outer sampling of NUM_SAMPLES_MATCHES cells per (pair, REINFORCE iteration) from final_scores (:136-137): the
    inference sampler, `mk_exprace_topk` (exponential race == torch.multinomial without replacement);
  * NUM_CORR_3d3d-point hypotheses and the <= NUM_REF_STEPS refinement rounds of all B*IT_MATCHES*IT_RANSAC hypotheses
    (:148-184, the no-grad block): `mk_train_ransac_masks`, one wave per hypothesis;
  * the REINFORCE scatter the reference runs as a python loop over B*IT_MATCHES rows (:251-261): `mk_reinforce_scatter`;
  * everything differentiable -- the gather + back-projection of the sampled keypoints (:139-146, `mk_gather_backproject` /
    `mk_gather_backproject_bwd`), the final masked Procrustes of every hypothesis with its soft inlier score and VCRE / pose loss
    (:187-227, `mk_train_tail_fwd` / `mk_train_tail_bwd`: one wave per hypothesis, the 3x3 SVD differentiated in closed form) and
    the softmax aggregation over the hypotheses of a set and the sets of a pair (:229-246, `mk_train_aggregate_fwd` / `_bwd`) --
    is ONE torch.autograd.Function (`_RansacLoss`) whose forward and backward are chains of HIP entry points;
  * what is left to torch: autograd's bookkeeping around that one node and the batch-level lines of `RANSAC_vectorized`
    (:287-333: means over B, the top-k curriculum mask), so that `avg_loss.backward()` fills
    outputs['kps0'|'kps1'|'depth0'|'depth1'].grad as the reference's trainer expects (lib/models/MicKey/model.py:101-128).

There is no CPU path: tensors must live on the GPU and the HIP library must be loadable.
"""
import torch

from . import ops
from ._native import MickeyHipError


class _RansacLoss(torch.autograd.Function):
    """reference loss_class.py:139-246 + :263-268 for all B * IT_MATCHES * IT_RANSAC hypotheses at once: gather + back-projection of
    the sampled matches (:139-146), hypotheses + refinement (no grad, :148-184), the masked Procrustes / soft inlier score / loss of
    every hypothesis (:187-227), the softmax aggregation over the hypotheses of a set and the sums over the sets of a pair
    (:229-246, :263-268).  Forward = five HIP entry points (mk_gather_backproject, mk_train_ransac_masks, mk_train_tail_fwd,
    mk_train_aggregate_fwd), backward = three (mk_train_aggregate_bwd, mk_train_tail_bwd, mk_gather_backproject_bwd): no ATen
    arithmetic between the sampled indices and dL/d(kps, depth).  Differentiable inputs: kps0, depth0, kps1, depth1; what the
    reference detaches (scores, mask, ground truth, intrinsics) gets no gradient."""

    @staticmethod
    def forward(ctx, kps0, depth0, kps1, depth1, scores, idx_outer, K0, K1, Rgt, tgt, Ko0, Ko1, cfg, idx_inner, rng):
        it_m, it_r, S, nc, th_ref, nref, th3d, loss_type, soft, temp, add_null, null_loss, null_score = cfg
        seed, offset, set_base = rng
        B, n0, n1 = scores.shape
        X, Y, w, corr = ops.gather_backproject(idx_outer, scores, kps0, depth0, kps1, depth1, K0, K1, it_m)
        mask, idx_in, rounds = ops.train_ransac_masks(X, Y, w, it_r, th_ref, nref, nc, idx_in=idx_inner, seed=seed, offset=offset,
                                                      set_base=set_base)
        out, Rt, saved = ops.train_tail_fwd(X, Y, mask, Rgt, tgt, Ko0, Ko1, it_r, it_m, th3d, loss_type, soft)
        loss_value, per_pair, coef, flags = ops.train_aggregate_fwd(out, Rt, saved, B, it_m, it_r, temp, add_null, null_loss, null_score)
        ctx.save_for_backward(X, Y, mask, Rgt, tgt, Ko0, Ko1, Rt, saved, coef, corr, idx_outer, K0, K1)
        ctx.cfg = (B, n0, n1, it_m, it_r, th3d, loss_type, soft)
        extras = (loss_value, out, mask, idx_in, rounds, Rt, saved, flags)
        # rotation / translation errors are LOGGING outputs (the reference logs them, model.py:151-171, and optimises avg_loss only):
        # returned without a gradient -- backpropagating avg_loss_rot / avg_loss_trans raises instead of yielding a partial gradient
        rot, trans = per_pair[:, 1].contiguous(), per_pair[:, 2].contiguous()
        ctx.mark_non_differentiable(rot, trans, *extras)
        return (per_pair[:, 0].contiguous(), rot, trans) + extras

    @staticmethod
    def backward(ctx, g_base, *unused):
        X, Y, mask, Rgt, tgt, Ko0, Ko1, Rt, saved, coef, corr, idx_outer, K0, K1 = ctx.saved_tensors
        B, n0, n1, it_m, it_r, th3d, loss_type, soft = ctx.cfg
        if g_base is None:   # the baseline did not take part in what was differentiated
            return (None,) * 15
        g = ops.train_aggregate_bwd(coef, g_base, B, it_m, it_r)
        gX, gY = ops.train_tail_bwd(X, Y, mask, Rgt, tgt, Ko0, Ko1, it_r, it_m, th3d, loss_type, soft, Rt, saved, g)
        gk0, gd0, gk1, gd1 = ops.gather_backproject_bwd(idx_outer, corr, gX, gY, K0, K1, B, it_m, n0, n1)
        return (gk0, gd0, gk1, gd1) + (None,) * 11


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
        scores = matches.float().contiguous()
        rowp = scores.view(B, ncell)
        outputs = {"kps0": kps0, "kps1": kps1, "depth0": depth0, "depth1": depth1}
        dbg = {}

        def bail():
            z = lambda *shape: torch.zeros(shape, device=dev)  # noqa: E731
            out = (z(B), z(B, 1), z(B, 1), z(B, ncell), z(B, ncell), outputs, 0)
            return out + (dbg,) if return_debug else out

        call = self._calls
        self._calls += 1
        pair_base = int(batch.get("pair_base", self._default_pair_base()))
        if idx_outer is None:
            # the sampler's own input scan raises `invalid` on NaN / inf / negative cells and on empty rows: the cases in
            # which the reference skips the loop (:118-124) or lands in its except branch (:263-270)
            invalid = torch.zeros((1,), device=dev, dtype=torch.int32)
            idx_outer, cnt = ops.exprace_topk(rowp, it_m, S, seed=self.seed, offset=2 * call, invalid=invalid, pair_base=pair_base)
            if int(invalid.item()) != 0 or int((cnt < S).any().item()) != 0:
                print("Invalid matching matrix! Skip RANSAC loop.")
                return bail()
        idx_outer = idx_outer.to(device=dev, dtype=torch.int32).contiguous()
        Rgt, tgt, K0, K1 = self.read_pose_parameters(batch)
        pair_b = lambda v, w: v.float().reshape(B, w).contiguous()  # noqa: E731
        vcre = self.loss_type == "VCRE"
        cfg = (it_m, it_r, S, nc, float(self.inlier_ref_th), self.num_ref_steps, float(self.inlier_3d_th), 0 if vcre else 1,
               bool(self.soft_clipping), float(self.score_temperature), bool(self.add_null_hypothesis), float(self.max_loss_null),
               float(self.th_outliers * S))
        baseline, losses_rot, losses_trans, loss_value, out_k, mask, idx_in, rounds, Rt, saved, flags = _RansacLoss.apply(
            kps0, depth0, kps1, depth1, scores, idx_outer, pair_b(K0, 9), pair_b(K1, 9), pair_b(Rgt, 9), pair_b(tgt, 3),
            pair_b(batch["Kori_color0"], 9) if vcre else None, pair_b(batch["Kori_color1"], 9) if vcre else None, cfg, idx_inner,
            (self.seed, 2 * call + 1, pair_base * it_m))
        if return_debug:
            dbg.update(idx_outer=idx_outer.long(), idx_inner=idx_in, inliers_final=mask, rounds=rounds, R=Rt[:, :9].reshape(Ri, 3, 3),
                       t=Rt[:, 9:].reshape(Ri, 1, 3))
        # the two conditions under which the reference drops the iteration, read back in ONE synchronisation: a rank-one
        # cross-covariance (torch.linalg.matrix_rank(H) == 1, :190, only when asked) and non-finite R / t (:225-227)
        nonfinite, rank_one = flags.tolist()
        if check_rank and rank_one > 0:
            print("[ERROR]: Skipping RANSAC iteration due to rank matrix.")
            return bail()
        if nonfinite:
            print("[ERROR]: Skipping RANSAC iteration due to invalid values in R/t.")
            return bail()
        gradients, gradients_b = ops.reinforce_scatter(idx_outer, loss_value, B, it_m, ncell)
        if return_debug:
            score_k = out_k[:, 3].reshape(Ro, it_r)
            if self.add_null_hypothesis:
                score_k = torch.cat([score_k, torch.full((Ro, 1), float(self.th_outliers * S), device=dev)], -1)
            dbg.update(loss_value=loss_value.reshape(Ro, 1), score=score_k)
        out = (baseline, losses_rot.unsqueeze(-1), losses_trans.unsqueeze(-1), gradients, gradients_b, outputs, 1)
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
