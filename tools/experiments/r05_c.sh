#!/bin/bash
# Round 5, GPU call C: where the transform launches spend their time now (phase stamps, per-kernel durations alone and under the overlap)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_c
mkdir -p $O
cd $R
L=$R/jxl-oxide_amd/csrc
export NZ=0.15
echo "=== phase stamps (prof lib, one stream)" | tee -a $O/log.txt
FRAMES=16 REPS=2 JXLGPU_LIB=$L/libjxlgpu_prof.so timeout 200 python tools/bench_transform.py "JXLGPU_NO_BATCH_OVERLAP=1" 2>&1 < /dev/null | grep -E "tr_prof|wall" | tee -a $O/log.txt
cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1
for mode in "JXLGPU_NO_BATCH_OVERLAP=1" ""; do
  tag=$( [ -n "$mode" ] && echo alone || echo overlap )
  rm -rf $O/stats_$tag
  FRAMES=32 REPS=4 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python $R/tools/bench_transform.py "$mode" > $O/stats_$tag.log 2>&1 < /dev/null
  f=$(find $O/stats_$tag -name "*kernel_stats.csv" | head -1)
  echo "=== kernel stats $tag" | tee -a $O/log.txt
  [ -n "$f" ] && cp "$f" $O/kernel_stats_$tag.csv && head -14 "$f" | cut -c1-150 | tee -a $O/log.txt
  rm -rf $O/stats_$tag
done
echo "r05_c done"
