#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 > gpurun_out/r02_pytest_gpu.txt
cat gpurun_out/r02_pytest_gpu.txt
timeout 300 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-h2d --no-single 2>/dev/null | tail -1 > gpurun_out/r02_bench_b1.json
python -c "
import json
d=json.load(open('gpurun_out/r02_bench_b1.json')); print(d['value'], d['ms_per_step']); [print(s['stage'], round(s['ms_per_step'],3)) for s in d['roofline']['stages']]"
