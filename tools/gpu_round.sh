#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "layernorm or norm" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof -o b32 -- python /root/repo/bench.py --no-cpu-baseline --no-kernel-events --steps 3 --warmup 1 > /root/repo/gpurun_out/prof_bench.log 2>&1
