#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof1 -o b1 -- python /root/repo/bench.py --batch 1 --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-events > /root/repo/gpurun_out/prof_b1.log 2>&1
tail -1 /root/repo/gpurun_out/prof_b1.log | cut -c1-200
