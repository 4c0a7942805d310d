timeout 60 tools/ubench/conv_in_bench
for mu in 64 96 64 96; do echo "== --max-units $mu"; timeout 600 python bench.py --max-units $mu --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg --no-roofline 2>/dev/null | cut -c1-200; done
