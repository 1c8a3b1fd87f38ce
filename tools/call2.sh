mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k flash_attention 2>&1 | tail -15 > gpurun_out/r03b_attn_tests.txt
cat gpurun_out/r03b_attn_tests.txt
timeout 300 python tools/bench_attn.py 2 4 7 2>&1 | tail -3 | tee gpurun_out/r03b_bench_attn.txt
timeout 1800 python -m pytest tests -q -m gpu --deselect tests/test_kernels_gpu.py 2>&1 | tail -15 > gpurun_out/r03b_pytest_rest.txt
cat gpurun_out/r03b_pytest_rest.txt
