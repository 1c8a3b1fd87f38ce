#!/bin/bash
# Verification + evidence round on the GPU box (gpurun): the GPU test suite, smoke(), the default bench line, a rocprofv3 kernel
# trace of the same command, the PMC passes (tools/profile_round.sh), the clock / matrix-pipe probe of the matrix kernels next to
# hipBLASLt (tools/pmc_clock.sh, ONLY_GEMM=1 for the short form), the GEMM shapes against hipBLASLt, the folded-LayerNorm forms,
# the matcher paths.  Outputs under gpurun_out/${TAG}_*; copy to profiles/.
export TAG=${TAG:-r05}
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then   # SKIP_TESTS=1: the suite and smoke() ran in a call of their own
  timeout 2400 python -m pytest tests -q -m gpu -rf 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.txt
  cat gpurun_out/${TAG}_pytest_gpu.txt
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.txt
fi
TAG=$TAG bash tools/profile_round.sh
TAG=$TAG bash tools/pmc_clock.sh 2>&1 | tail -50 > gpurun_out/${TAG}_pmc_clock.log
timeout 600 python tools/bench_gemm.py 7 -1 2>&1 | grep "M=" | tee gpurun_out/${TAG}_gemm_vs_hipblaslt.txt
timeout 600 python tools/bench_lnfold.py 2>&1 | tail -4 | tee gpurun_out/${TAG}_bench_lnfold.txt
timeout 300 python tools/bench_attn.py 1 2 3 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench_attn.txt
timeout 300 python tools/bench_matcher.py 2>&1 | tail -14 | tee gpurun_out/${TAG}_bench_matcher.txt
