#!/bin/bash
# ViT-S encoder GEMM shapes (K = 384 / 1536): which schedule?
mkdir -p gpurun_out
WIDTH=384 timeout 600 python tools/bench_gemm.py 1 2 7 -1 2>&1 | grep "M=" | tee gpurun_out/r04p_gemm_vits.txt
