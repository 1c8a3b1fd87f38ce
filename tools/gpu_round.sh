#!/bin/bash
# One GPU visit (rewritten per experiment during development).
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 > gpurun_out/r2_pytest_gpu.txt
cat gpurun_out/r2_pytest_gpu.txt
timeout 400 python bench.py --include-h2d 2>&1 | tail -1 > gpurun_out/r2_bench_b32.json
cat gpurun_out/r2_bench_b32.json
