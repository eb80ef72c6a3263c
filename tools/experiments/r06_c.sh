#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_vardct.py -m gpu -x -q 2>&1 | tail -30 | tee $O/vardct.txt
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -m gpu -x -q -k config2 2>&1 | tail -40 | tee $O/cfg2.txt
