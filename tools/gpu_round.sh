#!/bin/bash
# Verification + evidence round on the GPU box (gpurun): the GPU test suite, smoke(), a rocprofv3 kernel trace + the PMC passes of
# the default bench command and the default bench line itself (tools/profile_round.sh), the same bench with every leg
# (detail file only), the encoder GEMMs against hipBLASLt / persistent vs one tile per workgroup, the folded-LayerNorm forms, the
# attention variants, the matcher paths; PMC_CLOCK=1 adds the clock / matrix-pipe probe (tools/pmc_clock.sh, several minutes).
# Outputs under gpurun_out/${TAG}_*; copy to profiles/.
export TAG=${TAG:-r05}
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then   # SKIP_TESTS=1: the suite and smoke() ran in a call of their own
  timeout 2400 python -m pytest tests -q -m gpu -rf --durations=15 2>&1 | tail -40 > gpurun_out/${TAG}_pytest_gpu.txt
  tail -22 gpurun_out/${TAG}_pytest_gpu.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.txt
fi
TAG=$TAG bash tools/profile_round.sh
timeout 900 python bench.py --steps 10 --warmup 3 --legs all --sustained --precision --include-h2d --detail gpurun_out/${TAG}_bench_legs_detail.json \
  2>gpurun_out/${TAG}_bench_legs.err | tail -1 > gpurun_out/${TAG}_bench_legs_line.json
tail -12 gpurun_out/${TAG}_bench_legs.err
[ -n "$PMC_CLOCK" ] && TAG=$TAG bash tools/pmc_clock.sh 2>&1 | tail -50 > gpurun_out/${TAG}_pmc_clock.log
timeout 600 python tools/bench_persist.py 2>&1 | grep "M=" | tee gpurun_out/${TAG}_gemm_persistent.txt
timeout 600 python tools/bench_gemm.py 7 -1 2>&1 | grep "M=" | tee gpurun_out/${TAG}_gemm_vs_hipblaslt.txt
timeout 600 python tools/bench_lnfold.py 2>&1 | tail -4 | tee gpurun_out/${TAG}_bench_lnfold.txt
timeout 300 python tools/bench_attn.py 1 2 3 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench_attn.txt
timeout 300 python tools/bench_matcher.py 2>&1 | tail -14 | tee gpurun_out/${TAG}_bench_matcher.txt
