#!/bin/bash
# Round 6, GPU call V: (1) the fuzzer's finding of call T narrowed; (2) the D = 4 self-correcting predictor step
# (predict_lanes_wp4_kernel) against the oracle: the Modular suites + a fuzz minute; (3) tools/chain_probe; (4) the predictor pass
# timed with the old and the new step at 8K, and the old one at 4K / 2K (is the chain bound by its own issue or by its SIMD neighbours?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_v
mkdir -p $O
cd $R
timeout 300 python tools/experiments/r06_v_debug.py > $O/debug.txt 2>&1; tail -22 $O/debug.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_modular.py tests/test_gpu_schedules.py tests/test_gpu_baseline_sizes.py -m gpu -q -x 2>&1 | tail -5 | tee $O/tests.txt
timeout 200 python tests/tools/fuzz_parity.py 90 6201 2>&1 | tail -4 | cut -c1-500 | tee $O/fuzz.txt
tools/_bin/chain_probe | tee $O/chain_probe.txt
cd /tmp && export TMPDIR=/tmp
run() { # tag, env, size
  tag=$1; shift
  env $1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- python $R/tools/bench_modular.py $2 $3 > $O/bench_$tag.txt 2>&1
  echo "== $tag: $(tail -1 $O/bench_$tag.txt | cut -c1-200)"
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  grep -E "predict_" $f | awk -F'","' '{printf "   %-70s calls %s avg %s ns\n", substr($1,1,100), $2, $4}' | sed 's/(anonymous namespace):://g' | cut -c1-220
  cp $f $O/kernel_stats_$tag.csv; rm -rf $O/prof_$tag
}
run new_8k JXLGPU_X=0 7680 4320
run old_8k JXLGPU_PRED_STEP_V1=1 7680 4320
run old_4k JXLGPU_PRED_STEP_V1=1 3840 2160
run old_2k JXLGPU_PRED_STEP_V1=1 1920 1080
run new_2k JXLGPU_X=0 1920 1080
echo "r06_v done"
