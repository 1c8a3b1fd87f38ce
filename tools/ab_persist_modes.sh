# A/B of the persistent tile loop inside the whole forward: three alternating runs of bench.py --lean per mode (mk_gemm_set_tile 600 all /
# 602 producers only / 601 off) -> gpurun_out/r06u_persist_ab.txt (LABNOTES R6.9)
for rep in 1 2 3; do for m in 600 602 601; do python bench.py --lean --steps 10 --warmup 3 --gemm-tile $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode $m', round(d['value'],1), [(s['stage'], round(s['ms_per_step'],2)) for s in d['stages'][:2]])"; done; done | tee gpurun_out/r06u_persist_ab.txt
