#!/bin/bash
# round 4, GPU call D: matcher pass 2 through LDS, split convs writing planes, split grouped GEMM -- tests, matcher bench, bench legs
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest -q -m gpu -rf -s tests/test_kernels_gpu.py -k "softmax or matcher or split or golden" 2>&1 | grep -v "^$" | tail -14 | tee gpurun_out/r04d_pytest_kernels.txt
timeout 1200 python -m pytest -q -m gpu -rf tests/test_model_gpu.py tests/test_bench_config_gpu.py -k "split or golden or invariance or matcher or full_size or config5" 2>&1 | tail -8 | tee gpurun_out/r04d_pytest_model.txt
timeout 600 python tools/bench_matcher.py dual 2>&1 | tail -5 | tee gpurun_out/r04d_bench_matcher.txt
timeout 900 python bench.py --steps 10 --warmup 3 --legs fp16,ref_split --no-sustained --no-cpu-baseline --no-h2d --no-single 2>gpurun_out/r04d_bench.err | tail -1 > gpurun_out/r04d_bench_b32.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04d_bench_b32.json'))
print('value',d['value'],'ms',d['ms_per_step'])
for s in d['roofline']['stages']: print(s['stage'], round(s['ms_per_step'],2), s.get('achieved'), s.get('frac'))
for k,v in d.get('legs',{}).items():
    print(k, v.get('value'), v.get('ms_per_step'))
    for s in v.get('stages',[]): print('    ', s['stage'], round(s['ms_per_step'],2), s.get('achieved'))
PY
