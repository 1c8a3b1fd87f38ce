#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" 2>&1 | tail -4 | cut -c1-300
python tools/bench_kernels.py 2>&1 | grep flash
