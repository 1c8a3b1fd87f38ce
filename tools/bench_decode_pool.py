"""Host side of the input pipeline (row N1) at 8 ranks: N processes, each pinned to its share of the cores
(mickey_amd.distributed.affinity_plan, as bench.py --gpus N / a torchrun evaluation does) and each running the decode pool
PairFeeder runs (PIL decode on `workers` threads into a preallocated uint8 slot), against ONE machine's decode budget.

    python tools/bench_decode_pool.py --ranks 8 --seconds 3 [--size 540 720]

Prints one JSON line: frames/s per rank, their sum, and the same machine with one un-pinned process using the same total
number of threads -- the question it answers: does the feed collapse when 8 ranks decode at once?  (No GPU involved: the
pinned ring / H2D / preprocess stages need one and are measured by tools/bench_feeder.py.)"""
import argparse
import io
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_jpeg(w, h, seed):
    from PIL import Image
    g = np.random.default_rng(seed)
    base = g.integers(0, 232, ((h + 7) // 8, (w + 7) // 8, 3), dtype=np.uint8)
    img = np.kron(base, np.ones((8, 8, 1), dtype=np.uint8))[:h, :w] + g.integers(0, 24, (h, w, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img.astype(np.uint8)).save(buf, format="JPEG", quality=90)
    return buf.getvalue()


def worker(rank, ranks, seconds, w, h, pin, threads, q):
    import concurrent.futures
    from mickey_amd.distributed import affinity_plan
    from mickey_amd.input_pipeline import decode_rgb
    if pin:
        os.sched_setaffinity(0, affinity_plan(rank, ranks, sorted(os.sched_getaffinity(0))))
    ncpu = len(os.sched_getaffinity(0))
    nthreads = threads or max(2, ncpu // 2)              # PairFeeder's rule
    jpegs = [make_jpeg(w, h, 100 * rank + i) for i in range(4)]
    slots = [np.empty((h, w, 3), np.uint8) for _ in range(nthreads)]

    def decode(i):
        slots[i % nthreads][...] = decode_rgb(jpegs[i % 4])
        return 1
    n, t0 = 0, time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(nthreads) as pool:
        while time.perf_counter() - t0 < seconds:
            n += sum(pool.map(decode, range(n, n + 4 * nthreads)))
    q.put((rank, n / (time.perf_counter() - t0), ncpu, nthreads))


def run(ranks, seconds, w, h, pin=True, threads=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, ranks, seconds, w, h, pin, threads, q)) for r in range(ranks)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60 + 10 * seconds) for _ in ps)
    for p in ps:
        p.join(30)
    return res


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--size", type=int, nargs=2, default=[540, 720], help="W H of the stored frames")
    a = ap.parse_args(argv)
    w, h = a.size
    many = run(a.ranks, a.seconds, w, h)
    total_threads = sum(r[3] for r in many)
    one = run(1, a.seconds, w, h, pin=False, threads=total_threads)
    out = {"ranks": a.ranks, "frame": [w, h], "cores_allowed": len(os.sched_getaffinity(0)),
           "frames_per_s_per_rank": [round(r[1], 1) for r in many], "cpus_per_rank": [r[2] for r in many],
           "threads_per_rank": [r[3] for r in many], "frames_per_s_total": round(sum(r[1] for r in many), 1),
           "one_process_same_threads_frames_per_s": round(one[0][1], 1),
           "pairs_per_s_feedable": round(sum(r[1] for r in many) / 2, 1)}
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
