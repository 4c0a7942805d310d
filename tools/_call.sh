timeout 60 tools/ubench/conv_in_bench
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -s -m gpu -p no:cacheprovider -k "conv_in or (bench_width and bf16 and 128) or tiny_box" 2>&1 | grep -E "parity|passed|failed|Error|error" | cut -c1-300
