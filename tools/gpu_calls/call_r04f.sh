#!/bin/bash
# matcher pass 2: column chunks per row block (how many output rows the chip has in flight), with and without line-aligned pieces
mkdir -p gpurun_out
timeout 300 python tools/bench_matcher.py dual 2>&1 | tail -8 > gpurun_out/r04f_normal.txt
timeout 300 python -m pytest -q -m gpu tests/test_kernels_gpu.py -k "softmax or matcher" 2>&1 | tail -2
touch mickey_amd/csrc/mk_matcher.hip; MK_MATCHER_ALIGN_PROBE=1 python -m mickey_amd.build 2>&1 | grep -c matcher
MK_MATCHER_ALIGN_PROBE=1 timeout 300 python tools/bench_matcher.py dual 2>&1 | tail -8 > gpurun_out/r04f_probe.txt
echo normal; cat gpurun_out/r04f_normal.txt; echo "aligned-piece probe (wrong results, timing only)"; cat gpurun_out/r04f_probe.txt
