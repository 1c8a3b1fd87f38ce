#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
for v in "default:" "attn4:--attn-mode 4" "tile5:--gemm-tile 5"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 200 python bench.py --no-cpu-baseline $flags 2>&1 | tail -1 > gpurun_out/ab_$name.json
  python - <<PY
import json; d=json.load(open("gpurun_out/ab_$name.json")); print("$name", round(d["value"],1), "pairs/s", round(d["ms_per_step"],1), "ms", round(d["roofline"]["achieved"],1), "TF")
PY
done; done
