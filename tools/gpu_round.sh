#!/bin/bash
# One GPU visit (rewritten per experiment during development).
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -m gpu -k "float32 or fp32 or tiny or fp16" -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r2_pytest_fp32.txt
cat gpurun_out/r2_pytest_fp32.txt
timeout 600 python tools/bench_gemm.py 7 10 12 13 > gpurun_out/r2_bench_gemm.txt 2>&1
cat gpurun_out/r2_bench_gemm.txt
