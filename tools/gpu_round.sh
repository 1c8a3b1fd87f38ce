#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 1200 python bench.py > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-1500
timeout 600 python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_b1.log 2>&1
tail -1 gpurun_out/bench_b1.log | cut -c1-200
cd /tmp && rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r03 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/rocprof_bench.log 2>&1
cp /tmp/prof/r03_kernel_stats.csv $R/gpurun_out/r03_kernel_stats.csv
head -8 $R/gpurun_out/r03_kernel_stats.csv | cut -c1-160
