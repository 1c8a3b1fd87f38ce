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
    # global index of the shard's first pair: keys the samplers' Philox streams, so sharded == unsharded poses
    out["pair_base"] = int(data.get("pair_base", 0)) + lo
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


class PoseGatherer:
    """The one collective of the path, kept off the critical path: the all-gather of the packed poses
    is issued on a SIDE stream that waits only for the kernels that produced R, t, conf; the main stream goes on with
    the next batch and the result is waited for just before it is read.  On CPU tensors (gloo tests) it is synchronous."""

    def __init__(self, device=None):
        self.stream = torch.cuda.Stream(device) if device is not None and torch.device(device).type == "cuda" else None

    def submit(self, R, t, conf, sizes=None):
        if self.stream is None or not R.is_cuda:
            return (None, gather_poses(R, t, conf, sizes))
        packed = pack_poses(R, t, conf)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(R.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            out = gather_poses(*unpack_poses(packed), sizes=sizes)
            done = torch.cuda.Event()
            done.record(self.stream)
        packed.record_stream(self.stream)
        return (done, out)

    def wait(self, handle):
        done, out = handle
        if done is not None:
            main = torch.cuda.current_stream(out[0].device)
            main.wait_event(done)
            for o in out:   # allocated on the side stream, used on the main one from here on: tell the caching allocator, or
                o.record_stream(main)   # the next submit() may reuse the block while queued main-stream work still reads it
        return out


def _empty_poses(device):
    return (torch.zeros((0, 3, 3), device=device), torch.zeros((0, 1, 3), device=device), torch.zeros((0, 1), device=device))


def forward_local(model, local, sizes, gatherer=None):
    """The second half of forward_sharded for a caller that already holds only ITS slice of the global batch (`local`, with
    `pair_base` set; `sizes` = every rank's slice length): forward (skipped for an empty slice), then the all-gather."""
    if local["image0"].shape[0] > 0:
        R, t = model(local)
        conf = local["inliers"]
    else:
        R, t, conf = _empty_poses(local["image0"].device)
        local["R"], local["t"], local["inliers"] = R, t, conf
    if gatherer is not None:
        return gatherer.wait(gatherer.submit(R, t, conf, sizes))
    return gather_poses(R, t, conf, sizes)


def forward_sharded(model, data, return_local=False, gatherer=None):
    """Run model.forward on this rank's shard of `data` and all-gather the poses.  A rank whose shard is empty (global
    batch smaller than the world size, e.g. the ragged last batch of an evaluation) skips the model call but still
    takes part in the collective.  With a PoseGatherer the collective runs on its side stream."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    B = data["image0"].shape[0]
    local = shard_batch(data, rank, world)
    sizes = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    if sizes[rank] > 0:
        R, t = model(local)
        conf = local["inliers"]
    else:
        try:
            dev = next(model.parameters()).device
        except (StopIteration, AttributeError):
            dev = data["image0"].device
        R, t, conf = _empty_poses(dev)
        local["R"], local["t"], local["inliers"] = R, t, conf
    if gatherer is not None:
        Rg, tg, cg = gatherer.wait(gatherer.submit(R, t, conf, sizes))
    else:
        Rg, tg, cg = gather_poses(R, t, conf, sizes)
    return (Rg, tg, cg, local) if return_local else (Rg, tg, cg)


# ---- host-side placement of the ranks of one node -------------------------------------------------------------------
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/.../local_cpulist)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def affinity_plan(local_rank, local_world, avail_cpus, numa_cpus=None, numa_peers=None):
    """CPU set of one rank -- a pure function (unit-tested).  Every rank drives ~330 kernel launches per forward from
    Python and owns a decode pool; N ranks left on the scheduler's default mask share, and migrate over, all cores of both
    sockets.  Rule: the cores of the GPU's own NUMA node (`numa_cpus`, from sysfs) that this process may use, divided
    evenly among the `numa_peers` = (index of this rank among the local ranks on that node, their number); without NUMA
    information (node -1: VMs, containers) an even contiguous split of the available cores over the local ranks.  Never
    returns an empty set: a rank that would get nothing keeps the whole candidate set."""
    avail = sorted(avail_cpus)
    cand = [c for c in (numa_cpus or []) if c in set(avail)]
    if cand and numa_peers:
        idx, n = numa_peers
    else:
        cand, idx, n = avail, local_rank, local_world
    lo, hi = shard_range(len(cand), idx, max(1, n))
    return cand[lo:hi] or cand


def gpu_numa_cpus(device_index):
    """(numa_node, cpus of that node) of a visible GPU from sysfs, or (-1, None) when the platform does not say."""
    import os
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        node = int(open(os.path.join(base, "numa_node")).read())
        cpus = parse_cpulist(open(os.path.join(base, "local_cpulist")).read())
        return node, (cpus or None)
    except Exception:   # no sysfs entry, no such attribute in this torch build, not a PCI device ...: no pinning information
        return -1, None


def pin_rank(local_rank, local_world):
    """Pin this process as affinity_plan says and size torch's intra-op pool to the set.  sched_setaffinity(0, ...) moves
    only the calling thread and is inherited by threads created AFTERWARDS, so call this before init_process_group (RCCL
    proxy threads) where possible; threads that already exist (HIP runtime, torch pools) are moved one by one through
    /proc/self/task.  Returns what was done: {'numa_node', 'cpus', 'n_cpus', 'threads_pinned', 'threads_seen'} or {'error': ...}."""
    import os
    try:
        avail = sorted(os.sched_getaffinity(0))
        node, ncpus = gpu_numa_cpus(local_rank)
        peers = None
        if node >= 0 and ncpus:
            nodes = [gpu_numa_cpus(r)[0] for r in range(local_world)]
            same = [r for r in range(local_world) if nodes[r] == node]
            peers = (same.index(local_rank), len(same))
        cpus = affinity_plan(local_rank, local_world, avail, ncpus, peers)
        os.sched_setaffinity(0, cpus)
        seen = pinned = 0
        try:
            tids = [int(t) for t in os.listdir("/proc/self/task")]
        except OSError:
            tids = []
        for tid in tids:
            seen += 1
            try:
                os.sched_setaffinity(tid, cpus)
                pinned += 1
            except OSError:   # the thread ended meanwhile
                pass
        torch.set_num_threads(max(1, min(len(cpus), 16)))
        return {"numa_node": node, "cpus": "%d-%d" % (cpus[0], cpus[-1]) if cpus == list(range(cpus[0], cpus[-1] + 1)) else cpus,
                "n_cpus": len(cpus), "threads_pinned": pinned, "threads_seen": seen}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}
