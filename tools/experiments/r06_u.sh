#!/bin/bash
# Round 6, GPU call U: is the self-correcting predictor pass bound by its chain's latency or by issue slots shared between
# waves of one SIMD?  (a) tools/chain_probe: latency of dependent instructions and LDS round trips at 1 / 2 / 4 waves per SIMD;
# (b) the Modular stage at 8K, 4K and 2K under the kernel trace: the longest chain is 1276 steps at every size, the number of
# waves per SIMD is 1.5 / 0.37 / 0.1.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_u
mkdir -p $O
cd $R
tools/_bin/chain_probe | tee $O/chain_probe.txt
cd /tmp && export TMPDIR=/tmp
for sz in "7680 4320" "3840 2160" "1920 1080"; do
  tag=$(echo $sz | tr ' ' x)
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o p -- python $R/tools/bench_modular.py $sz > $O/bench_$tag.txt 2>&1
  tail -1 $O/bench_$tag.txt | cut -c1-300
  f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1)
  grep -E "predict_" $f | awk -F'","' '{printf "   %-60s calls %s avg %s ns\n", substr($1,1,90), $2, $4}' | sed 's/(anonymous namespace):://g' | cut -c1-200
  cp $f $O/kernel_stats_$tag.csv; rm -rf $O/prof_$tag
done
echo "r06_u done"
