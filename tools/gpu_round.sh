#!/bin/bash
# one GPU visit: pp64 GEMM correctness + A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or conv3x3" 2>&1 | tail -5
timeout 300 python tools/bench_gemm.py 2 3 5 2>&1 | tail -10
timeout 200 python bench.py --no-cpu-baseline --gemm-tile 6 2>&1 | tail -1
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1
