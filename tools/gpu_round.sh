#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "gemm or qkv or residual or patch" 2>&1 | tail -4
timeout 400 python tools/bench_gemm.py 7 9 2>&1 | tail -4
