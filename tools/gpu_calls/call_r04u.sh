#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/bench_conv_order.py 2>&1 | grep "order" | tee gpurun_out/r04u_conv_order.txt
