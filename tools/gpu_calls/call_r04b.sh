#!/bin/bash
# round 4, GPU call B: the new kernels (split-fp16 matcher, split-operand convs) -- their tests, everything they touch
# (whole-forward, batch invariance, branches), then the bench line with the fp16 / ref_split legs and a kernel trace
mkdir -p gpurun_out
timeout 1200 python -m pytest -q -m gpu -rf tests/test_kernels_gpu.py -k "softmax or matcher or split or conv3x3" 2>&1 | tail -30 > gpurun_out/r04b_pytest_kernels.txt
tail -12 gpurun_out/r04b_pytest_kernels.txt
timeout 1500 python -m pytest -q -m gpu -rf -s tests/test_model_gpu.py tests/test_config_branches_gpu.py tests/test_bench_config_gpu.py tests/test_callers_gpu.py 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r04b_pytest_model.txt
tail -40 gpurun_out/r04b_pytest_model.txt
timeout 900 python bench.py --steps 10 --warmup 3 --legs fp16,ref_split --no-sustained --no-cpu-baseline --no-h2d 2>gpurun_out/r04b_bench.err | tail -1 > gpurun_out/r04b_bench_b32.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04b_bench_b32.json'))
print('value',d['value'],'ms',d['ms_per_step'])
for s in d['roofline']['stages']: print(s['stage'], round(s['ms_per_step'],2), s.get('achieved'), s.get('frac'))
for k,v in d.get('legs',{}).items():
    print(k, v.get('value'), v.get('ms_per_step'))
    for s in v.get('stages',[]): print('    ', s['stage'], round(s['ms_per_step'],2), s.get('achieved'))
print('single',d.get('single_pair'))
PY
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04b_prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --lean > $GRAFT_REPO_ROOT/gpurun_out/r04b_prof.log 2>&1
cd $GRAFT_REPO_ROOT && f=$(ls gpurun_out/r04b_prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -40 "$f" > gpurun_out/r04b_kernel_stats.csv && head -25 gpurun_out/r04b_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/r04b_prof
