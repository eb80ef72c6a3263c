#!/bin/bash
# Round 6, GPU call AI: fewer predictor waves resident per CU (JXLGPU_PRED_LDS_PAD), one and three contexts
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_ai
mkdir -p $O
cd $R
one() { # tag, env
  env $2 timeout 400 python bench.py --config 3 --frames 12 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/$1.json 2> $O/err.txt
  echo "$1 [$2]: $(python -c "import json,sys; d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"
}
one c3_pad0 "JXLGPU_BENCH_CONTEXTS=3"
one c3_pad6k "JXLGPU_BENCH_CONTEXTS=3 JXLGPU_PRED_LDS_PAD=6144"
one c3_pad9k "JXLGPU_BENCH_CONTEXTS=3 JXLGPU_PRED_LDS_PAD=9216"
one c3_pad14k "JXLGPU_BENCH_CONTEXTS=3 JXLGPU_PRED_LDS_PAD=14336"
one c3_pad28k "JXLGPU_BENCH_CONTEXTS=3 JXLGPU_PRED_LDS_PAD=28672"
one c1_pad0 "JXLGPU_BENCH_CONTEXTS=1"
one c1_pad6k "JXLGPU_BENCH_CONTEXTS=1 JXLGPU_PRED_LDS_PAD=6144"
one c5_pad6k "JXLGPU_BENCH_CONTEXTS=5 JXLGPU_PRED_LDS_PAD=6144"
one c5_pad14k "JXLGPU_BENCH_CONTEXTS=5 JXLGPU_PRED_LDS_PAD=14336"
echo "r06_ai done"
