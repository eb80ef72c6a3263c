#!/bin/bash
# Round 6, GPU call X: config 3 (8K Modular, self-correcting predictor residuals) with the round-3 predictor step and the D = 4 one, A/B/A/B,
# then the kernel trace of both.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_x
mkdir -p $O
cd $R
for v in 0 1 0 1; do
  JXLGPU_PRED_STEP_V1=$v timeout 400 python bench.py --config 3 --frames 8 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/cfg3_v1_$v.json 2> $O/cfg3.err
  echo "PRED_STEP_V1=$v: $(python -c "import json,sys; d=json.loads([l for l in open('$O/cfg3_v1_$v.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"; tail -1 $O/cfg3.err | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  JXLGPU_PRED_STEP_V1=$v timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o p -- python $R/bench.py --config 3 --frames 8 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify < /dev/null > $O/prof_$v.log 2>&1
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    echo "-- kernel stats, PRED_STEP_V1=$v"
    head -8 "$f" | awk -F'","' '{printf "   %-80s calls %s avg %s ns  %s %%\n", substr($1,1,110), $2, $4, $5}' | sed 's/(anonymous namespace):://g' | cut -c1-230
    cp "$f" $O/kernel_stats_v1_$v.csv
  fi
  rm -rf $O/prof_$v
done
echo "r06_x done"
