# A/B of attention modes inside the whole forward: alternating bench.py --lean runs (mk_attn_set_mode 2 = default with the one-sub-block
# tail wave, 6 = without it) -> gpurun_out/${TAG:-r06w}_attn_ab.txt
for rep in 1 2 3; do for m in 2 6; do python bench.py --lean --steps 10 --warmup 3 --attn-mode $m 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn mode $m', round(d['value'],1), [(s['stage'], round(s['ms_per_step'],2)) for s in d['stages'][:2]])"; done; done | tee gpurun_out/${TAG:-r06w}_attn_ab.txt
