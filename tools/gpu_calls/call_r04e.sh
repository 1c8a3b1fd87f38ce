#!/bin/bash
# round 4, GPU call E: matcher pass 2 with chunk-major workgroup order
mkdir -p gpurun_out
timeout 600 python -m pytest -q -m gpu -rf tests/test_kernels_gpu.py tests/test_bench_config_gpu.py -k "softmax or matcher or invariance" 2>&1 | tail -4 | tee gpurun_out/r04e_pytest.txt
timeout 600 python tools/bench_matcher.py dual 2>&1 | tail -5 | tee gpurun_out/r04e_bench_matcher.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o p -- python $GRAFT_REPO_ROOT/tools/bench_matcher.py dual > /tmp/prof_m.log 2>&1
f=$(ls /tmp/prof_m/*kernel_stats.csv /tmp/prof_m/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r04e_matcher_kernel_stats.csv && python - <<'PY'
import csv, os
for r in csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04e_matcher_kernel_stats.csv")):
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", "min", round(float(r["MinNs"]) / 1e3, 1), "max", round(float(r["MaxNs"]) / 1e3, 1))
PY
