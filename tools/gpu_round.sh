#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or conv3x3 or qkv or patch" 2>&1 | tail -2
timeout 300 python tools/bench_gemm.py 7 5 2>&1 | tail -8
timeout 300 python tools/bench_epilogues.py 2>&1 | tail -4
timeout 300 python tools/gemm_timeline.py 5 2>&1 | tail -3
