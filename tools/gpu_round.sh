#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 > gpurun_out/r02_pytest_gpu.txt
cat gpurun_out/r02_pytest_gpu.txt
bash tools/profile_round.sh
