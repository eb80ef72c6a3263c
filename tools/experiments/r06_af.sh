#!/bin/bash
# Round 6, GPU call AF: config 3, frames spread over C contexts — sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_af
mkdir -p $O
cd $R
one() { # tag, env, frames
  env $2 timeout 400 python bench.py --config 3 --frames $3 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/$1.json 2> $O/err.txt
  echo "$1 [$2 frames=$3]: $(python -c "import json,sys; d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"
}
one c3 JXLGPU_BENCH_CONTEXTS=3 8
one c4 JXLGPU_BENCH_CONTEXTS=4 8
one c5 JXLGPU_BENCH_CONTEXTS=5 8
one c8 JXLGPU_BENCH_CONTEXTS=8 8
one c1_12 JXLGPU_BENCH_CONTEXTS=1 12
one c3_12 JXLGPU_BENCH_CONTEXTS=3 12
one c4_12 JXLGPU_BENCH_CONTEXTS=4 12
one c6_12 JXLGPU_BENCH_CONTEXTS=6 12
one c3_4 JXLGPU_BENCH_CONTEXTS=3 4
one c4_4 JXLGPU_BENCH_CONTEXTS=4 4
one c3_12_p1 "JXLGPU_BENCH_CONTEXTS=3 JXLGPU_PRED_PERSIST=1" 12
one c3_12_v1 "JXLGPU_BENCH_CONTEXTS=3 JXLGPU_PRED_STEP_V1=1" 12
echo "r06_af done"
