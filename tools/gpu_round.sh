#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 1200 python bench.py > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-1700
timeout 600 python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_b1.log 2>&1
tail -1 gpurun_out/bench_b1.log | cut -c1-200
