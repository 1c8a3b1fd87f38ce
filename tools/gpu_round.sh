#!/bin/bash
mkdir -p gpurun_out
timeout 300 python bench.py --dtype fp16 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_fp16.json
python - <<PY
import json
d=json.load(open("gpurun_out/bench_fp16.json")); print("fp16", round(d["value"],1), "pairs/s", round(d["ms_per_step"],2), "ms", round(d["roofline"]["achieved"],1), "TF", d["finite_output"])
PY
