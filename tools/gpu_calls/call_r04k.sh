#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_bench_config_gpu.py -q -m gpu -x 2>&1 | tail -12 | tee gpurun_out/r04k_pytest_solver.txt
timeout 300 python tools/bench_sampler.py 2>&1 | grep "B=" | tee gpurun_out/r04k_bench_sampler.txt
B=1 timeout 300 python tools/bench_sampler.py 2>&1 | grep "B=" | tee -a gpurun_out/r04k_bench_sampler.txt
cd /tmp
rm -rf /tmp/prof_k
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o p -- python $R/tools/bench_sampler.py > /tmp/prof_k.log 2>&1
python - <<'PY' | tee $R/gpurun_out/r04k_sampler_kernel_stats.txt
import csv, glob
for r in csv.DictReader(open(glob.glob("/tmp/prof_k/*kernel_stats.csv")[0])):
    if "exprace" in r["Name"] or "zero_u32" in r["Name"]:
        print("%-60s calls %5s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cd $R
timeout 600 python bench.py --steps 10 --warmup 3 --lean 2>gpurun_out/r04k_bench.err | tail -1 > gpurun_out/r04k_bench_lean.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04k_bench_lean.json"))
print("value", d["value"], "ms", d["ms_per_step"])
for s in d["roofline"]["stages"]:
    print(s["stage"], round(s["ms_per_step"], 3), s.get("achieved"))
PY
