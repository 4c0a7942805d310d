for v in 1 0 1; do echo "== IDF_EPI_STAGE=$v"; timeout 100 tools/ubench/big_sched_stage$v 5 3; done
