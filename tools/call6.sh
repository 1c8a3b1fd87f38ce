mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash_attention and (10- or 11-)" 2>&1 | tail -4
timeout 300 python tools/bench_attn.py 9 10 11 2 2>&1 | tail -1 | tee gpurun_out/r03e_bench_attn.txt
for m in 9 11; do
timeout 300 python bench.py --lean --steps 10 --warmup 3 --attn-mode $m 2>gpurun_out/bench_err_$m.txt | tail -1 > gpurun_out/r03_bench_attn_mode$m.json
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r03_bench_attn_mode$m.json"))
    print("attn-mode $m:", round(d["value"],1), "pairs/s", [ (s["stage"], round(s["ms_per_step"],2)) for s in d["roofline"]["stages"][:3]])
except Exception as e:
    print("mode $m failed", e); print(open("gpurun_out/bench_err_$m.txt").read()[-1500:])
PY
done
