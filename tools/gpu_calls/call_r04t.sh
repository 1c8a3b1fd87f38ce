#!/bin/bash
# bench.py after its legs were made failure-tolerant: the default invocation once more (the line of record stays the evidence call's)
mkdir -p gpurun_out
timeout 900 python bench.py 2>gpurun_out/r04t_bench.err | tail -1 > gpurun_out/r04t_bench_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04t_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["traffic"], sorted(d["legs"]), [k for k in d if k not in ("legs", "roofline", "config")])
print({k: v.get("value", v.get("error")) for k, v in d["legs"].items()})
PY
