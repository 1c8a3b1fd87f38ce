#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/prof_s
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-h2d --no-single > /tmp/prof_s.log 2>&1
cp $(ls /tmp/prof_s/*kernel_stats.csv | head -1) $R/gpurun_out/r2_small_stats.csv
