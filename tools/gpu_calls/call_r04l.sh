#!/bin/bash
# end of round 4: the default bench line once more (with the per-role traffic factors in place) and the per-kernel comparisons
# on the final sources
export TAG=r04
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 3 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_b32.json
head -c 600 gpurun_out/${TAG}_bench_b32.json; echo
timeout 600 python tools/bench_gemm.py 7 -1 2>&1 | grep "M=" | tee gpurun_out/${TAG}_gemm_vs_hipblaslt.txt
timeout 600 python tools/bench_lnfold.py 2>&1 | tail -4 | tee gpurun_out/${TAG}_bench_lnfold.txt
timeout 300 python tools/bench_attn.py 1 2 3 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench_attn.txt
timeout 300 python tools/bench_matcher.py 2>&1 | tail -14 | tee gpurun_out/${TAG}_bench_matcher.txt
