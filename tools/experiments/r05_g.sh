#!/bin/bash
# Round 5, GPU call G: hardware queues (GPU_MAX_HW_QUEUES): do the ctx's streams share queues?
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_g
mkdir -p $O
cd $R
L=$R/jxl-oxide_amd/csrc
export NZ=0.15 FRAMES=32 REPS=6
for q in 2 4 8 16; do
  echo "=== GPU_MAX_HW_QUEUES=$q" | tee -a $O/sweep.log
  GPU_MAX_HW_QUEUES=$q JXLGPU_LIB=$L/libjxlgpu.so timeout 300 python tools/bench_transform.py "" "JXLGPU_BATCH_STREAM_ROWS=96" 2>&1 < /dev/null | grep -v "^CANARY" | tee -a $O/sweep.log
done
cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1
GPU_MAX_HW_QUEUES=8 FRAMES=32 REPS=4 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/bench_transform.py "" > $O/stats.log 2>&1 < /dev/null
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_overlap_q8.csv && head -12 "$f" | cut -c1-150
rm -rf $O/stats
echo "r05_g done"
