#!/bin/bash
# Round 6, GPU call AK: inverse-Squeeze segment length (JXLGPU_SQZ_SEG) with one and three contexts
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_ak
mkdir -p $O
cd $R
one() { # tag, env
  env $2 timeout 400 python bench.py --config 3 --frames 12 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/$1.json 2> $O/err.txt
  echo "$1 [$2]: $(python -c "import json,sys; d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"
}
for seg in 64 32 48 96 128; do
  one c3_seg$seg "JXLGPU_BENCH_CONTEXTS=3 JXLGPU_SQZ_SEG=$seg"
done
for seg in 64 32 128; do
  one c1_seg$seg "JXLGPU_BENCH_CONTEXTS=1 JXLGPU_SQZ_SEG=$seg"
done
echo "r06_ak done"
