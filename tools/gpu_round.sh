#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_solver_gpu.py tests/test_bench_config_gpu.py -q -x -m gpu 2>&1 | tail -4
timeout 600 python -m pytest tests/test_model_gpu.py -q -x -m gpu -k "planted or determinism or graph" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-alt --no-h2d 2>&1 | tail -1 > gpurun_out/r2_bench_sampler.json; python -c "
import json
d=json.load(open('gpurun_out/r2_bench_sampler.json')); print(d['value'], d['ms_per_step']); [print(s['stage'], round(s['ms_per_step'],2), s.get('achieved')) for s in d['roofline']['stages']]"
