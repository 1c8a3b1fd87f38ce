#!/bin/bash
# round 4, GPU call A: the whole GPU suite (all failures reported, no -x), smoke(), the keypoint bisect, the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -rf 2>&1 | tail -60 > gpurun_out/r04a_pytest_gpu.txt
tail -25 gpurun_out/r04a_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r04a_smoke.txt
timeout 300 python tools/diag_kps.py 2>&1 | tail -12 | tee gpurun_out/r04a_diag_kps.txt
timeout 900 python bench.py --steps 20 --warmup 3 2>gpurun_out/r04a_bench.err | tail -1 > gpurun_out/r04a_bench_b32.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_bench_b32.json'))
print('value',d['value'],'ms',d['ms_per_step'])
for s in d['roofline']['stages']: print(s['stage'], round(s['ms_per_step'],2), s.get('achieved'), s.get('frac'))
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('legs',{}).items()})
print('single',d.get('single_pair')); print('prec',d.get('precision'))
PY
