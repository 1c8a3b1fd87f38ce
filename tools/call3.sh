mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash_attention and 8-" 2>&1 | tail -8 > gpurun_out/r03c_attn_tests.txt
cat gpurun_out/r03c_attn_tests.txt
timeout 300 python tools/bench_attn.py 2 4 7 8 2>&1 | tail -2 | tee gpurun_out/r03c_bench_attn.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "ln_fold" -s 2>&1 | grep -E "centring|passed|failed|Error|assert" | tail -40 > gpurun_out/r03c_ln_tests.txt
cat gpurun_out/r03c_ln_tests.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_config_gpu.py -q 2>&1 | tail -8 | tee gpurun_out/r03c_model_tests.txt
