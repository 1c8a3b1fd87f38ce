mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r03f_pytest_gpu.txt
cat gpurun_out/r03f_pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 3 2>gpurun_out/r03f_bench.err | tail -1 > gpurun_out/r03f_bench_b32.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03f_bench_b32.json"))
print(d["value"], d["ms_per_step"], "sustained", d.get("sustained",{}).get("value"), "single", d.get("single_pair",{}).get("ms_per_pair"))
for s in d["roofline"]["stages"]: print(s["stage"], round(s["ms_per_step"],3), s.get("achieved"), s.get("frac"))
for k,v in d.get("legs",{}).items(): print(k, round(v["value"],1), v["ms_per_step"])
PY
