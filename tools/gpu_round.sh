#!/bin/bash
# One GPU visit (rewritten per experiment during development).  This version: verification of the committed state --
# full GPU tests, smoke, rocprofv3 kernel statistics of bench.py, bench lines at B = 32 and B = 1 into gpurun_out/.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -3 > gpurun_out/r01_pytest_gpu.txt
cat gpurun_out/r01_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof -o b32 -- python /root/repo/bench.py --no-cpu-baseline --no-kernel-events --steps 3 --warmup 1 > /root/repo/gpurun_out/prof_bench.log 2>&1
cd /root/repo
timeout 300 python bench.py 2>&1 | tail -1 > gpurun_out/bench_b32.json
timeout 300 python bench.py --batch 1 --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_b1.json
