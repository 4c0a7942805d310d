for v in 3 1 3 1; do echo "== IDF_MLP_V=$v"; IDF_MLP_V=$v timeout 120 tools/ubench/mlp_harness 5 2>&1 | grep -E "^\[|fused|M [0-9]"; done
