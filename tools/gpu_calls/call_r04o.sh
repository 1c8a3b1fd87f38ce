#!/bin/bash
# the GPU suite and smoke() once more on the final tree (tests added after the evidence call)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -rf 2>&1 | tail -15 > gpurun_out/r04_pytest_gpu.txt
cat gpurun_out/r04_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r04_smoke.txt
