#!/bin/bash
# Verification round on the GPU box: the GPU test suite, then the evidence of tools/profile_round.sh.
TAG=${TAG:-r03}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.txt
cat gpurun_out/${TAG}_pytest_gpu.txt
TAG=$TAG bash tools/profile_round.sh
