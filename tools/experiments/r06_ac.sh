#!/bin/bash
# Round 6, GPU call AC: the predictor pass as a fixed number of workgroups per SIMD over a queue of waves (JXLGPU_PRED_PERSIST=W)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_ac
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_modular.py tests/test_gpu_baseline_sizes.py -m gpu -q -x < /dev/null 2>&1 | tail -3 | tee $O/tests.txt
JXLGPU_PRED_PERSIST=1 timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py -m gpu -q -x -k "config3 or modular" < /dev/null 2>&1 | tail -3 | tee $O/tests_w1.txt
one() { # tag, env
  env $2 timeout 400 python bench.py --config 3 --frames 8 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/$1.json 2> $O/err.txt
  echo "$1 [$2]: $(python -c "import json,sys; d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"
}
one w0 JXLGPU_PRED_PERSIST=0
one w1 JXLGPU_PRED_PERSIST=1
one w2 JXLGPU_PRED_PERSIST=2
one w3 JXLGPU_PRED_PERSIST=3
one w0b JXLGPU_PRED_PERSIST=0
one w2b JXLGPU_PRED_PERSIST=2
one w1_late0 "JXLGPU_PRED_PERSIST=1 JXLGPU_PRED_LATE_STEPS=0"
one w2_late0 "JXLGPU_PRED_PERSIST=2 JXLGPU_PRED_LATE_STEPS=0"
cd /tmp && export TMPDIR=/tmp
for w in 1 2; do
  JXLGPU_PRED_PERSIST=$w timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -o p -- python $R/bench.py --config 3 --frames 8 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify < /dev/null > $O/prof_$w.log 2>&1
  f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then echo "-- W=$w"; grep predict_lanes_wp4 "$f" | awk -F'",' '{print $2}' | cut -c1-70; cp "$f" $O/kernel_stats_w$w.csv; fi
  rm -rf $O/prof_$w
done
echo "r06_ac done"
