#!/bin/bash
# One GPU visit (rewritten per experiment during development).
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_mapfree_eval_gpu.py tests/test_train_ransac_gpu.py -q -x -m gpu 2>&1 | tail -8
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 > gpurun_out/r2_pytest_gpu.txt
cat gpurun_out/r2_pytest_gpu.txt
timeout 600 python tools/bench_train_ransac.py 2>&1 | tail -1 > gpurun_out/r2_bench_train_ransac.json
cat gpurun_out/r2_bench_train_ransac.json
