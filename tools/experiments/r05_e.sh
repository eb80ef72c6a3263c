#!/bin/bash
# Round 5, GPU call E: heavy transform launches on their own stream (not behind the ring launch of the previous chunk)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_e
mkdir -p $O
cd $R
L=$R/jxl-oxide_amd/csrc
export NZ=0.15 FRAMES=32 REPS=6
run() { lib=$1; shift; echo "=== lib $lib" | tee -a $O/sweep.log; JXLGPU_LIB=$L/$lib timeout 300 python tools/bench_transform.py "$@" 2>&1 < /dev/null | grep -v "^CANARY" | tee -a $O/sweep.log; }
run libjxlgpu.so "" "JXLGPU_NO_BATCH_OVERLAP=1" "JXLGPU_TR_SIDE_MAX=0" "JXLGPU_BATCH_HEAVY=24" "JXLGPU_BATCH_HEAVY=8" "JXLGPU_BATCH_STREAM_ROWS=1072" "JXLGPU_BATCH_STREAM_ROWS=716" "JXLGPU_BATCH_STREAM_ROWS=268" "JXLGPU_BATCH_CHUNK=8" "JXLGPU_BATCH_CHUNK=12" "JXLGPU_STREAM_PRIO=1" "JXLGPU_STREAM_PRIO=-1"
cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1
FRAMES=32 REPS=4 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/bench_transform.py "" > $O/stats.log 2>&1 < /dev/null
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_overlap.csv && head -12 "$f" | cut -c1-150
rm -rf $O/stats
echo "r05_e done"
