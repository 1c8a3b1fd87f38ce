#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/bench_two_streams.py 20 2>&1 | grep "B=" | tee gpurun_out/r04m_two_streams.txt
B=8 timeout 300 python tools/bench_two_streams.py 40 2>&1 | grep "B=" | tee -a gpurun_out/r04m_two_streams.txt
