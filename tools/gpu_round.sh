#!/bin/bash
MK_ATTN_ABLATIONS=1 timeout 300 python tools/bench_kernels.py 2>&1 | grep flash | grep "nimg': 64"
