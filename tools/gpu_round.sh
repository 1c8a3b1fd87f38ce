#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_config_gpu.py -q -x -m gpu -k "matcher or softmax or dual or sinkhorn or mutual or invariance or bench_config" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -m gpu 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-alt --no-h2d 2>&1 | tail -1 > gpurun_out/r2_bench_matcher.json; python -c "
import json
d=json.load(open('gpurun_out/r2_bench_matcher.json')); print(d['value'], d['ms_per_step']); [print(s['stage'], round(s['ms_per_step'],2), s.get('achieved')) for s in d['roofline']['stages']]"
