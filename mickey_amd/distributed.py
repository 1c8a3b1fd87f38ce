"""Multi-GPU harness: image pairs are independent (no cross-pair op anywhere in forward; BatchNorm is
in eval mode), so a global batch is sharded contiguously over ranks -- one process per GPU -- and the
ONLY exchange is one all-gather of a packed [B_local, 13] fp32 tensor per batch (R row-major 9, t 3,
confidence 1) over RCCL/xGMI (52 B per pair: latency-bound).  The reference itself is
single-process (submission.py); this is new, thin plumbing on torch.distributed."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(data, rank, world):
    """Slice every batched tensor / list of a data dict to this rank's pairs."""
    B = data["image0"].shape[0]
    lo, hi = shard_range(B, rank, world)
    out = {}
    for k, v in data.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B:
            out[k] = v[lo:hi]
        elif isinstance(v, (list, tuple)) and len(v) == B:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def pack_poses(R, t, conf):
    B = R.shape[0]
    return torch.cat([R.reshape(B, 9), t.reshape(B, 3), conf.reshape(B, 1)], 1).float().contiguous()


def unpack_poses(p):
    B = p.shape[0]
    return p[:, :9].reshape(B, 3, 3), p[:, 9:12].reshape(B, 1, 3), p[:, 12:13]


def gather_poses(R, t, conf, sizes=None):
    """All ranks receive the poses of the whole global batch, in global pair order.  `sizes` = per-rank
    shard sizes when they differ (ragged last batch); equal shards take the single-collective path."""
    packed = pack_poses(R, t, conf)
    if not (dist.is_available() and dist.is_initialized()):
        return unpack_poses(packed)
    world = dist.get_world_size()
    if sizes is None or len(set(sizes)) == 1:
        out = packed.new_empty((world * packed.shape[0], 13))
        dist.all_gather_into_tensor(out, packed)
        return unpack_poses(out)
    mx = max(sizes)
    pad = packed.new_zeros((mx, 13))
    pad[: packed.shape[0]] = packed
    out = packed.new_empty((world * mx, 13))
    dist.all_gather_into_tensor(out, pad)
    keep = torch.cat([out[r * mx: r * mx + s] for r, s in enumerate(sizes)], 0)
    return unpack_poses(keep)


def forward_sharded(model, data, return_local=False):
    """Run model.forward on this rank's shard of `data` and all-gather the poses."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    B = data["image0"].shape[0]
    local = shard_batch(data, rank, world)
    R, t = model(local)
    sizes = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    Rg, tg, cg = gather_poses(R, t, local["inliers"], sizes)
    return (Rg, tg, cg, local) if return_local else (Rg, tg, cg)
