#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_solver_gpu.py -q -x 2>&1 | tail -3
timeout 300 python bench.py --batch 1 --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b1_auto.json
python - <<PY
import json
for f in ("b1_auto",):
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, round(d["value"],1), "pairs/s", round(d["ms_per_step"],2), "ms", "graph", d["config"].get("hip_graph"))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
