#!/bin/bash
timeout 300 python tools/bench_gemm.py 408 404 416 432 402 2>&1 | tail -5
