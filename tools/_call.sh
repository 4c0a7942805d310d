for r in 1 2; do timeout 120 tools/ubench/mlp_harness 5 2>&1 | grep -E "^\[|fused|M [0-9]"; done
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -m gpu -p no:cacheprovider -k "fused_mlp" 2>&1 | grep -E "parity|passed|failed|Error|error" | cut -c1-300
