#!/bin/bash
# Round 6, GPU call AE: config 3 with consecutive frames on two / three contexts, now that the predictor pass is issue-bound
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_ae
mkdir -p $O
cd $R
one() { # tag, env
  env $2 timeout 400 python bench.py --config 3 --frames 8 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/$1.json 2> $O/err.txt
  echo "$1 [$2]: $(python -c "import json,sys; d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"
}
one c1 JXLGPU_BENCH_CONTEXTS=1
one c2 JXLGPU_BENCH_CONTEXTS=2
one c3 JXLGPU_BENCH_CONTEXTS=3
one c2b JXLGPU_BENCH_CONTEXTS=2
one c2p "JXLGPU_BENCH_CONTEXTS=2 JXLGPU_PRED_PERSIST=1"
echo "r06_ae done"
