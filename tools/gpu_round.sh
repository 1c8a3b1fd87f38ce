#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -m gpu -k "linear_attention or golden or invariance or full_size" 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-alt --no-h2d --no-single 2>&1 | tail -1 > gpurun_out/r2_bench_x.json; python -c "
import json
d=json.load(open('gpurun_out/r2_bench_x.json')); print(d['value'], d['ms_per_step']); [print(s['stage'], round(s['ms_per_step'],2), s.get('achieved'), s.get('frac')) for s in d['roofline']['stages']]"
