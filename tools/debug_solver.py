import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mickey_amd import ops, synthetic as syn
from mickey_amd.config import default_cfg
from oracle import mickey_oracle as O
dev = torch.device("cuda:0")
cfg = default_cfg(); scfg = copy.deepcopy(cfg); scfg.PROCRUSTES.IT_MATCHES = 4; scfg.PROCRUSTES.IT_RANSAC = 25
data, _, _ = syn.planted_pose_problem(B=3, h=14, w=12, seed=4321, angle_deg=(1.0, 1.5), t_norm=(0.03, 0.04))
torch.manual_seed(5)
Ro, to, co, dbg = O.estimate_pose({k: v.clone() for k, v in data.items()}, scfg, return_debug=True)
X, Y, w = dbg["X"].to(dev), dbg["Y"].to(dev), dbg["weights"].to(dev)
for it_r, sub in ((25, slice(None)), (2, slice(0, 24))):
    i3 = dbg["idx3"].int().reshape(12, 25, 3)[:, :it_r].reshape(-1, 3).contiguous()
    Rh, th, sc, _ = ops.ransac_hypotheses(X, Y, w, it_r, 0.3, idx3_in=i3.to(dev))
    ref = dbg["R_hyp"].reshape(12, 25, 3, 3)[:, :it_r].reshape(-1, 3, 3)
    refs = dbg["score"].reshape(12, 25)[:, :it_r].reshape(-1)
    e = (Rh.cpu().reshape(-1, 3, 3) - ref).norm(dim=(1, 2))
    print("it_r", it_r, "err", e[:12], "score err", (sc.cpu() - refs).abs()[:12])
    print("Rh0", Rh[0].cpu(), "ref0", ref[0])
# smaller k
for k in (64, 512, 1024, 2048):
    i3 = (dbg["idx3"] % k).int()
    Rh, th, sc, _ = ops.ransac_hypotheses(X[:, :k].contiguous(), Y[:, :k].contiguous(), w[:, :k].contiguous(), 25, 0.3, idx3_in=i3.to(dev))
    gsel = torch.arange(12).repeat_interleave(25)[:, None].expand(-1, 3)
    Rr, tr, H = O.kabsch(dbg["X"][:, :k][gsel, i3.long()], dbg["Y"][:, :k][gsel, i3.long()])
    e = (Rh.cpu().reshape(-1, 3, 3) - Rr).norm(dim=(1, 2))
    print("k", k, "median err", e.median().item(), "frac>1e-3", (e > 1e-3).float().mean().item())
