#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_solver_gpu.py -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof -o b32 -- python /root/repo/bench.py --no-cpu-baseline --no-kernel-events --steps 3 --warmup 1 > /root/repo/gpurun_out/prof_bench.log 2>&1
tail -1 /root/repo/gpurun_out/prof_bench.log | cut -c1-100
