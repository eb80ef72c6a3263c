#!/bin/bash
# Round 6, GPU call M: whole -m gpu suite on the final tree + the config 5 bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_m
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_suite.txt
timeout 900 python bench.py --config 5 --cpu-seconds 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err < /dev/null; echo "cfg5 rc=$?"; cut -c1-300 $O/bench_cfg5.json; tail -2 $O/bench_cfg5.err
echo "r06_m done"
