#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash_attention" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_bench_config_gpu.py tests/test_model_gpu.py -q -x 2>&1 | tail -3
timeout 300 python tools/bench_attn.py 1 2 3 2>&1 | tail -1
timeout 600 python bench.py --steps 10 --warmup 3 --lean > gpurun_out/r03i_bench.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03i_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for s in d['roofline']['stages'][:3]: print(s['stage'], round(s['ms_per_step'],3), round(s.get('achieved') or 0,1))
PY
