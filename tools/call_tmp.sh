#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "layernorm or linattn or linear_att or heads or posenc" > gpurun_out/r03f_tests.txt 2>&1
tail -3 gpurun_out/r03f_tests.txt
timeout 900 python -m pytest tests/test_model_gpu.py -q -x > gpurun_out/r03f_model_tests.txt 2>&1
tail -3 gpurun_out/r03f_model_tests.txt
timeout 600 python bench.py --steps 10 --warmup 3 --lean --no-legs --no-sustained > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03f_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for s in d['roofline']['stages']: print(s['stage'], round(s['ms_per_step'],3), round(s.get('achieved') or 0,1))
PY
