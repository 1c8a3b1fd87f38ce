#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
for t in 0 4; do
  echo "gemm-tile $t:" $(python bench.py --steps 4 --warmup 2 --no-cpu-baseline --gemm-tile $t 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])")
done
done
for a in 2 4; do
  echo "attn-mode $a:" $(python bench.py --steps 4 --warmup 2 --no-cpu-baseline --attn-mode $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
done
