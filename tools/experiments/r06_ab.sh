#!/bin/bash
# Round 6, GPU call AB: config 3 with issue priority by chain length and with snake launch orders of the predictor waves
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_ab
mkdir -p $O
cd $R
one() { # tag, env
  env $2 timeout 400 python bench.py --config 3 --frames 8 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/$1.json 2> $O/err.txt
  echo "$1 [$2]: $(python -c "import json,sys; d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"
}
one base JXLGPU_X=0
one prio JXLGPU_PRED_PRIO=1
one snake1024 JXLGPU_PRED_SNAKE=1024
one snake512 JXLGPU_PRED_SNAKE=512
one snake2048 JXLGPU_PRED_SNAKE=2048
one base2 JXLGPU_X=0
one prio_snake "JXLGPU_PRED_PRIO=1 JXLGPU_PRED_SNAKE=1024"
one late0 JXLGPU_PRED_LATE_STEPS=0
echo "r06_ab done"
