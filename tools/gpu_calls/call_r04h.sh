#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --steps 10 --warmup 3 --legs fp16,ref_split --no-sustained --no-cpu-baseline --no-h2d 2>gpurun_out/r04h_bench.err | tail -1 > gpurun_out/r04h_bench_b32.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h_bench_b32.json'))
print('value',d['value'],'ms',d['ms_per_step'])
for s in d['roofline']['stages']: print(s['stage'], round(s['ms_per_step'],2), s.get('achieved'), s.get('frac'))
for k,v in d.get('legs',{}).items():
    print(k, v.get('value'), v.get('ms_per_step'))
    for s in v.get('stages',[]): print('    ', s['stage'], round(s['ms_per_step'],2), s.get('achieved'))
print('single',d.get('single_pair'))
PY
