#!/bin/bash
# One GPU-box visit: kernel parity tests + micro-bench (+ optional extra command).  Logs -> gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/bench_kernels.py --json gpurun_out/kernels.json > gpurun_out/bench_kernels.log 2>&1
tail -5 gpurun_out/pytest_gpu.log
tail -25 gpurun_out/bench_kernels.log
