#!/bin/bash
# Round 6, GPU call S: config 3 with frames alternating between two contexts (consecutive frames overlap) against one context
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s
mkdir -p $O
cd $R
for c in 2 1 2 1 3; do
  JXLGPU_BENCH_CONTEXTS=$c timeout 400 python bench.py --config 3 --frames 8 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/cfg3_c$c.json 2> $O/cfg3.err; echo "contexts=$c: $(cut -c1-20,60-170 $O/cfg3_c$c.json)"; tail -1 $O/cfg3.err
done
echo "r06_s done"
