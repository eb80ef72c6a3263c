#!/bin/bash
# Round 6, GPU call AD: the row-start block of the D = 4 predictor step without its inner branch (rings of zeros for row 0): parity, config 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_ad
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_modular.py tests/test_gpu_baseline_sizes.py tests/test_gpu_schedules.py -m gpu -q -x < /dev/null 2>&1 | tail -3 | tee $O/tests.txt
one() { # tag, env
  env $2 timeout 400 python bench.py --config 3 --frames 8 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/$1.json 2> $O/err.txt
  echo "$1 [$2]: $(python -c "import json,sys; d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"
}
one a JXLGPU_X=0
one v1 JXLGPU_PRED_STEP_V1=1
one b JXLGPU_X=0
timeout 200 python tests/tools/fuzz_parity.py 100 6301 < /dev/null 2>&1 | tail -3 | cut -c1-400 | tee $O/fuzz.txt
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --config 3 --frames 8 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify < /dev/null > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep predict_lanes_wp4 "$f" | awk -F'",' '{print $2}' | cut -c1-70; cp "$f" $O/kernel_stats.csv; fi
rm -rf $O/prof
echo "r06_ad done"
