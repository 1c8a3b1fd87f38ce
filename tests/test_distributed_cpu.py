"""CPU, world_size 2, gloo: the sharding + single all-gather of the multi-GPU path (what runs over
RCCL/xGMI on the GPU node)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel:
    """Stands in for the GPU module: pose = deterministic function of the pair's image mean."""

    def __call__(self, data):
        B = data["image0"].shape[0]
        m = data["image0"].reshape(B, -1).mean(1)
        R = torch.eye(3).repeat(B, 1, 1) * m.view(B, 1, 1)
        t = torch.stack([m, 2 * m, 3 * m], 1).view(B, 1, 3)
        data["inliers"] = (10 * m).view(B, 1)
        data["R"], data["t"] = R, t
        return R, t


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mickey_amd import distributed as D
    g = torch.Generator().manual_seed(0)
    data = {"image0": torch.rand((B, 3, 4, 4), generator=g), "image1": torch.rand((B, 3, 4, 4), generator=g),
            "scene_id": ["s%d" % i for i in range(B)], "down": 14}
    R, t, c, local = D.forward_sharded(_FakeModel(), data, return_local=True)
    lo, hi = D.shard_range(B, rank, world)
    ok = local["scene_id"] == data["scene_id"][lo:hi] and local["down"] == 14
    q.put((rank, R, t, c, ok))
    dist.barrier()
    dist.destroy_process_group()


def _run(B, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    img = torch.rand((B, 3, 4, 4), generator=g)
    m = img.reshape(B, -1).mean(1)
    for rank, R, t, c, ok in res:
        assert ok
        assert R.shape == (B, 3, 3) and t.shape == (B, 1, 3) and c.shape == (B, 1)
        assert torch.allclose(R[:, 0, 0], m) and torch.allclose(t[:, 0, 2], 3 * m) and torch.allclose(c[:, 0], 10 * m)


def test_even_shards_single_allgather():
    _run(6)


def test_ragged_last_batch():
    _run(5)


def test_fewer_pairs_than_ranks():
    """B = 1 on 2 ranks: the rank with the empty shard skips the model call but still joins the collective (a
    hang here is what submission_io.predict(sharded=True) would have hit on the last batch of an evaluation)."""
    _run(1)


def test_world_8_ragged_and_sparse_batches():
    """The node the driver scales to: 8 ranks.  A global batch of 12 gives shards of 2,2,2,2,1,1,1,1 (the padded
    all-gather path), 16 the single-collective path, 5 leaves three ranks with an empty shard."""
    _run(12, world=8)
    _run(16, world=8)
    _run(5, world=8)


def test_affinity_plan_rules():
    from mickey_amd.distributed import affinity_plan, parse_cpulist
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == []
    # no NUMA information: an even contiguous split of what the process may use
    plans = [affinity_plan(r, 8, range(64)) for r in range(8)]
    assert plans[0] == list(range(0, 8)) and plans[7] == list(range(56, 64))
    assert sorted(c for p in plans for c in p) == list(range(64))
    # two sockets, GPUs 0-3 on node 0 (cpus 0-47, 96-143), 4-7 on node 1: each rank a quarter of ITS node, no overlap
    n0, n1 = parse_cpulist("0-47,96-143"), parse_cpulist("48-95,144-191")
    plans = [affinity_plan(r, 8, range(192), n0 if r < 4 else n1, (r % 4, 4)) for r in range(8)]
    assert all(len(p) == 24 for p in plans) and sorted(c for p in plans for c in p) == list(range(192))
    assert set(plans[0]) <= set(n0) and set(plans[5]) <= set(n1)
    # a cgroup that allows only part of the node: the plan stays inside it; more ranks than cores: nobody gets an empty set
    assert affinity_plan(1, 2, [4, 5, 6, 7], n0, (1, 2)) == [6, 7]
    assert all(affinity_plan(r, 8, [0, 1, 2]) for r in range(8))
    # NUMA cores that the process may not use at all -> fall back to the even split
    assert affinity_plan(0, 2, [200, 201], n0, (0, 2)) == [200]


def test_pose_gatherer_cpu_path_is_synchronous():
    from mickey_amd import distributed as D
    R, t, c = torch.eye(3).repeat(2, 1, 1), torch.ones(2, 1, 3), torch.full((2, 1), 7.0)
    g = D.PoseGatherer(None)
    Rg, tg, cg = g.wait(g.submit(R, t, c))
    assert torch.equal(Rg, R) and torch.equal(tg, t) and torch.equal(cg, c)


def test_shard_range_covers_everything():
    from mickey_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 33):
        for w in (1, 2, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


# ---- sharded evaluation: every rank feeds only ITS slice of each global batch (mapfree_eval.predict_to_zip) -------------
class _FakeFeeder:
    """Stands in for input_pipeline.PairFeeder (which needs a GPU): records carry a scalar `val`, the 'image' is that value;
    the records a rank was asked to feed are logged."""
    fed = []

    def __init__(self, records, batch_size, resize, device="cpu", batches=None, **kw):
        assert records is None and batches is not None and all(len(b) <= batch_size for b in batches)
        self.batches = batches

    def __iter__(self):
        for b in self.batches:
            _FakeFeeder.fed.extend(r["val"] for r in b)
            v = torch.tensor([r["val"] for r in b], dtype=torch.float32)
            yield {"image0": v.view(-1, 1, 1, 1).expand(-1, 3, 2, 2).contiguous(), "image1": torch.zeros(len(b), 3, 2, 2)}


def _eval_worker(rank, world, port, n, bs, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mickey_amd import input_pipeline, mapfree_eval as ME
    input_pipeline.PairFeeder = _FakeFeeder
    recs = [{"val": float(i + 1), "scene_id": "s%d" % (i % 2), "pair_names": ("a.jpg", "q%03d.jpg" % i)} for i in range(n)]
    seen = []

    class M(_FakeModel):
        def __call__(self, data):
            seen.append(int(data.get("pair_base", -1)))
            return super().__call__(data)
    res = ME.predict_to_zip(M(), recs, bs, (2, 2), os.path.join(tmp, "r%d.zip" % rank), sharded=True, device="cpu")
    lines = sorted(str(p) for plist in res.values() for p in plist)
    q.put((rank, list(_FakeFeeder.fed), seen, lines, os.path.exists(os.path.join(tmp, "r%d.zip" % rank))))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluation_feeds_only_the_local_slice(tmp_path):
    """7 records, global batches of 4 (4 + 3), 2 ranks: rank 0 feeds records {1,2,5,6}, rank 1 {3,4,7}; both end up with the
    poses of all 7 pairs in global order; pair_base = the slice's offset inside its global batch; only rank 0 writes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, 7, 4, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, fed0, seen0, lines0, wrote0), (r1, fed1, seen1, lines1, wrote1) = res
    assert fed0 == [1.0, 2.0, 5.0, 6.0] and fed1 == [3.0, 4.0, 7.0]
    assert seen0 == [0, 0] and seen1 == [2, 2]
    assert lines0 == lines1 and len(lines0) == 7
    assert wrote0 and not wrote1


def test_eight_ranks_decoding_at_once_do_not_collapse():
    """Row N1 at the node scale (tools/bench_decode_pool.py): 8 processes, each pinned to its share of the allowed cores and
    running PairFeeder's decode pool, against one machine's decode budget -- no rank starves and together they are not
    slower than one process with the same number of threads (which the GIL and core migration hold back)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import bench_decode_pool as BD
    many = BD.run(8, 1.0, 270, 360)
    rates = [r[1] for r in many]
    assert len(rates) == 8 and min(rates) > 0.25 * max(rates), rates
    one = BD.run(1, 1.0, 270, 360, pin=False, threads=sum(r[3] for r in many))
    assert sum(rates) > 0.5 * one[0][1], (rates, one)
