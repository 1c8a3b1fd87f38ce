#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -4
timeout 200 python bench.py --no-cpu-baseline --gemm-tile 4 2>&1 | tail -1 > gpurun_out/bench_tile4.json
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_tile0.json
timeout 200 python bench.py --no-cpu-baseline --gemm-tile 4 2>&1 | tail -1 > gpurun_out/bench_tile4b.json
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_tile0b.json
for f in tile4 tile0 tile4b tile0b; do python - <<PY
import json; d=json.load(open("gpurun_out/bench_$f.json")); print("$f", round(d["value"],1), "pairs/s", round(d["roofline"]["achieved"],1), "TF")
PY
done
