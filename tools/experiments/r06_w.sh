#!/bin/bash
# Round 6, GPU call W: call V again without its hang (rocprofv3 writes a rocpd database unless asked for csv; an empty file name made
# grep read stdin): the small-level chain fix and the D = 4 predictor step against the oracle, then the predictor pass timed with the
# round-3 step and the new one at 8K, and the old one at 2K.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_w
mkdir -p $O
cd $R
timeout 300 python tools/experiments/r06_v_debug.py < /dev/null > $O/debug.txt 2>&1; cut -c1-110 $O/debug.txt | tail -20
timeout 900 python -m pytest tests/test_gpu_modular.py tests/test_gpu_schedules.py tests/test_gpu_baseline_sizes.py -m gpu -q -x < /dev/null 2>&1 | tail -5 | tee $O/tests.txt
timeout 200 python tests/tools/fuzz_parity.py 60 6202 < /dev/null 2>&1 | tail -4 | cut -c1-500 | tee $O/fuzz.txt
cd /tmp && export TMPDIR=/tmp
run() { # tag, env, w, h
  tag=$1
  env $2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o p -- python $R/tools/bench_modular.py $3 $4 < /dev/null > $O/bench_$tag.txt 2>&1
  echo "== $tag: $(grep -h workload $O/bench_$tag.txt | cut -c1-200)"
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    grep -E "predict_" "$f" | awk -F'","' '{printf "   %-70s calls %s avg %s ns\n", substr($1,1,100), $2, $4}' | sed 's/(anonymous namespace):://g' | cut -c1-220
    cp "$f" $O/kernel_stats_$tag.csv
  fi
  rm -rf $O/prof_$tag
}
run new_8k JXLGPU_X=0 7680 4320
run old_8k JXLGPU_PRED_STEP_V1=1 7680 4320
run new_8k_b JXLGPU_X=0 7680 4320
run old_2k JXLGPU_PRED_STEP_V1=1 1920 1080
run new_2k JXLGPU_X=0 1920 1080
echo "r06_w done"
