#!/bin/bash
mkdir -p gpurun_out
S=$(date +%s)
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/r2_bench_default.json
echo "default bench took $(( $(date +%s) - S )) s"
python -c "
import json
d=json.load(open('gpurun_out/r2_bench_default.json')); print(d['value'], d['ms_per_step'], d.get('single_pair'), d.get('pcie_inclusive',{}).get('value'), d.get('alt',{}).get('value'), d['cpu_baseline']['value'])"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
timeout 120 python bench.py --gpus 2 --steps 1 --warmup 0 2>&1 | tail -2 | cut -c1-300; echo "rc=$?"
