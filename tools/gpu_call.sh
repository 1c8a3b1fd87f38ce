#!/bin/bash
# One development call on the GPU box: `gpurun -- 'TAG=r05a bash tools/gpu_call.sh <step> ...'`; every step writes under
# gpurun_out/${TAG}_*.  Steps: tests:<pytest args>, bench[:<bench args>], matcher, sampler, gemm, attn, lnfold
export TAG=${TAG:-r06x}
mkdir -p gpurun_out
for step in "$@"; do
  name=${step%%:*}
  arg=""
  [ "$name" != "$step" ] && arg=${step#*:}
  case $name in
    tests)   eval "timeout 1500 python -m pytest $arg -q -m gpu -x -rf" 2>&1 | tail -25 | tee gpurun_out/${TAG}_pytest_$(echo "$arg" | md5sum | cut -c1-6).txt ;;
    bench)   SECONDS=0; timeout 900 python bench.py $arg 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_b32.json
             cp gpurun_out/bench_detail.json gpurun_out/${TAG}_bench_detail.json 2>/dev/null
             echo "bench wall ${SECONDS}s" | tee gpurun_out/${TAG}_bench.time; wc -c gpurun_out/${TAG}_bench_b32.json; cat gpurun_out/${TAG}_bench_b32.json; tail -12 gpurun_out/${TAG}_bench.err ;;
    matcher) timeout 300 python tools/bench_matcher.py dual 2>&1 | tail -12 | tee gpurun_out/${TAG}_bench_matcher.txt ;;
    sampler) timeout 300 python tools/bench_sampler.py 2>&1 | tail -12 | tee gpurun_out/${TAG}_bench_sampler.txt ;;
    gemm)    timeout 600 python tools/bench_gemm.py $arg 2>&1 | grep "M=" | tee gpurun_out/${TAG}_gemm_vs_hipblaslt.txt ;;
    gemmb1)  timeout 600 python tools/bench_gemm_b1.py 2>&1 | grep "M=" | tee gpurun_out/${TAG}_gemm_b1.txt ;;
    train)   timeout 600 python tools/bench_train_ransac.py 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench_train_ransac.json ;;
    persist) timeout 600 python tools/bench_persist.py 2>&1 | grep "M=" | tee gpurun_out/${TAG}_gemm_persistent.txt ;;
    lnfold)  timeout 600 python tools/bench_lnfold.py 2>&1 | tail -6 | tee gpurun_out/${TAG}_bench_lnfold.txt ;;
    attn)    timeout 300 python tools/bench_attn.py $arg 2>&1 | tail -4 | tee gpurun_out/${TAG}_bench_attn.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
