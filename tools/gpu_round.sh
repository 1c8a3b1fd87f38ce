#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (+ rocprof).  Logs -> gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | tail -200 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_b8.log 2>&1
timeout 1200 python bench.py > gpurun_out/bench.log 2>&1
tail -40 gpurun_out/pytest_gpu.log
tail -5 gpurun_out/smoke.log
tail -3 gpurun_out/bench_b8.log
tail -3 gpurun_out/bench.log
