echo "== bench, pair hoist on (default)"
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg --no-roofline 2>/dev/null | cut -c1-260
echo "== bench, IDF_PAIR_HOIST=0"
IDF_PAIR_HOIST=0 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg --no-roofline 2>/dev/null | cut -c1-260
echo "== parity"
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_samplers_gpu.py -q -s -m gpu -p no:cacheprovider -k "not s50 or headline" 2>&1 | grep -E "parity|passed|failed|Error|error" | cut -c1-330
