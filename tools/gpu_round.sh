#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_solver_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/pytest_solver.log
tail -25 gpurun_out/pytest_solver.log | cut -c1-300
