#!/bin/bash
mkdir -p gpurun_out
for b in 1 4 8; do for t in 0 7; do
  timeout 200 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --gemm-tile $t 2>&1 | tail -1 > gpurun_out/ab.json
  python - <<PY
import json; d=json.load(open("gpurun_out/ab.json")); print("B=$b tile=$t", round(d["value"],1), "pairs/s", round(d["ms_per_step"],2), "ms", round(d["roofline"]["achieved"],1), "TF")
PY
done; done
