#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "tile_orders" 2>&1 | tail -3 | tee gpurun_out/r04n_pytest_tile_orders.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/micro/mfma_power.hip 2>/dev/null
timeout 120 /tmp/mfma_power | tee gpurun_out/r04n_mfma_power.txt
