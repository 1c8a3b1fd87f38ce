mkdir -p gpurun_out
bash tools/pmc_clock.sh 2>&1 | tail -45
for m in 2 7 8; do
timeout 300 python bench.py --lean --steps 10 --warmup 3 --attn-mode $m 2>/dev/null | tail -1 > gpurun_out/r03_bench_attn_mode$m.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03_bench_attn_mode$m.json"))
print("attn-mode $m:", round(d["value"],1), "pairs/s", [ (s["stage"], round(s["ms_per_step"],2)) for s in d["roofline"]["stages"][:3]])
PY
done
