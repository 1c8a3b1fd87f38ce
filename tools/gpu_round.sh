#!/bin/bash
timeout 300 python tools/bench_gemm.py 7 35 39 36 37 38 2>&1 | tail -6
