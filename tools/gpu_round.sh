#!/bin/bash
timeout 1200 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
timeout 300 python tools/bench_epilogues.py 2>&1 | tail -2
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_a.json
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_b.json
for f in a b; do python - <<PY
import json; d=json.load(open("gpurun_out/bench_$f.json")); print("$f", round(d["value"],1), "pairs/s", round(d["roofline"]["achieved"],1), "TF", round(d["ms_per_step"],1), "ms")
PY
done
