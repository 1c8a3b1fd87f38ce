#!/bin/bash
# round 4, GPU call C: producer-epilogue diet (tests + per-GEMM cost), matcher per-kernel split (exact vs split-fp16 path), Sinkhorn grouping
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -m gpu -rf tests/test_kernels_gpu.py -k "ln_fold or recentre or gemm_ls or gemm_bias or sinkhorn or golden" 2>&1 | tail -8 | tee gpurun_out/r04c_pytest_kernels.txt
timeout 900 python -m pytest -q -m gpu -rf tests/test_bench_config_gpu.py tests/test_model_gpu.py -k "invariance or config5 or golden" 2>&1 | tail -8 | tee gpurun_out/r04c_pytest_model.txt
timeout 600 python tools/bench_lnfold.py 2>&1 | tail -6 | tee gpurun_out/r04c_bench_lnfold.txt
timeout 600 python tools/bench_matcher.py 2>&1 | tail -12 | tee gpurun_out/r04c_bench_matcher.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o p -- python $R/tools/bench_matcher.py dual > /tmp/prof_m.log 2>&1
f=$(ls /tmp/prof_m/*kernel_stats.csv /tmp/prof_m/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r04c_matcher_kernel_stats.csv && head -14 "$f" | cut -c1-220
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o p -- python $R/bench.py --steps 5 --warmup 2 --lean > /tmp/prof_b.log 2>&1
tail -1 /tmp/prof_b.log | cut -c1-400 > $R/gpurun_out/r04c_bench_under_rocprof.txt
f=$(ls /tmp/prof_b/*kernel_stats.csv /tmp/prof_b/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r04c_bench_b32_kernel_stats.csv && head -30 "$f" | cut -c1-200
cd $R && timeout 600 python bench.py --steps 20 --warmup 3 --lean 2>/dev/null | tail -1 > gpurun_out/r04c_bench_lean.json; python -c "
import json; d=json.load(open('gpurun_out/r04c_bench_lean.json')); print('value',d['value'],'ms',d['ms_per_step']); [print(s['stage'], round(s['ms_per_step'],2), s.get('achieved')) for s in d['roofline']['stages']]"
