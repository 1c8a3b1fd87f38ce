#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "ls_residual or patch_embed or dual_softmax" 2>&1 | tail -4
