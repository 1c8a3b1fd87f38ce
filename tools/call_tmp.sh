#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_config_gpu.py -q -x -k "solver or hypoth or ransac or sampler or invariance" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_train_ransac_gpu.py -q -x 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --lean > /tmp/prof_x.log 2>&1
grep "^{\"metric\"" /tmp/prof_x.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
grep -i "hypoth" $(ls /tmp/prof_x/*kernel_stats.csv | head -1) | awk -F'",' '{print $2}' 
