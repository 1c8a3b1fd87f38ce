#!/bin/bash
# One GPU visit (rewritten per experiment during development).
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_input_pipeline_gpu.py -q -x -m gpu -k "attention or pipeline or resize or conv" 2>&1 | tail -8 > gpurun_out/r2_pytest_attn.txt
cat gpurun_out/r2_pytest_attn.txt
timeout 300 python tools/bench_attn.py 2 4 6 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-alt --attn-mode 6 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print(s['stage'], round(s['ms_per_step'],2), s.get('achieved')) for s in d['roofline']['stages']]"
