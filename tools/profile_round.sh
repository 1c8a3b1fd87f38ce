#!/bin/bash
# Round profile: the default bench line, a rocprofv3 kernel trace of the same command (shorter), B=1 line, and the PMC
# traffic passes.  Outputs under gpurun_out/ (copied to profiles/ by hand after inspection).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r02_bench_b32.json
timeout 300 python bench.py --batch 1 --no-cpu-baseline --no-alt --no-h2d --no-single 2>/dev/null | tail -1 > gpurun_out/r02_bench_b1.json
cd /tmp
rm -rf /tmp/prof_r02
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r02 -o p -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-h2d --no-single > /tmp/prof_r02.log 2>&1
grep "^{\"metric\"" /tmp/prof_r02.log | tail -1 > $R/gpurun_out/r02_bench_b32_under_rocprof.json
cp $(ls /tmp/prof_r02/*kernel_stats.csv | head -1) $R/gpurun_out/r02_bench_b32_kernel_stats.csv
bash $R/tools/pmc_bench_traffic.sh > /tmp/pmc_traffic.log 2>&1; tail -2 /tmp/pmc_traffic.log
head -c 600 $R/gpurun_out/r02_bench_b32.json; echo; head -5 $R/gpurun_out/r02_bench_b32_kernel_stats.csv | cut -c1-220
