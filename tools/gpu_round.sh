#!/bin/bash
# One GPU visit (rewritten per experiment during development).
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_ransac_gpu.py -q -x -m gpu -s 2>&1 | tail -25 > gpurun_out/r2_pytest_train.txt
cat gpurun_out/r2_pytest_train.txt
