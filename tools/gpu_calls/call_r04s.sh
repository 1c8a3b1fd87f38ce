#!/bin/bash
# attention on the 16x16x32 MFMA (modes 4 / 5) against the 32x32x16 kernel (1 / 2): parity, then time (32 pairs and one pair)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "flash_attention" 2>&1 | tail -2 | tee gpurun_out/r04s_pytest_attn.txt
timeout 300 python tools/bench_attn.py 1 2 4 5 2>&1 | tail -1 | tee gpurun_out/r04s_bench_attn.txt
NIMG=2 timeout 300 python tools/bench_attn.py 1 2 4 5 2>&1 | tail -1 | tee -a gpurun_out/r04s_bench_attn.txt
NIMG=8 timeout 300 python tools/bench_attn.py 1 2 4 5 2>&1 | tail -1 | tee -a gpurun_out/r04s_bench_attn.txt
