#!/bin/bash
# The rocprofv3 invocations behind profiles/ (run on the GPU box from the repo root, e.g. through
# gpurun).  Counters are collected in their own passes with --kernel-trace only.
#   tools/prof.sh stats   -> per-kernel durations of the headline bench
#   tools/prof.sh hbm     -> FETCH_SIZE / WRITE_SIZE (two passes), 2 frames, one stream
#   tools/prof.sh sq      -> SQ issue / wait counters
#   tools/prof.sh pk      -> the same SQ counters with the packed streaming kernel (JXLGPU_STREAM_PK=${PK:-1})
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_SHORT="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames-per-gpu 2 --streams 1"
case "$1" in
  stats) timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline ;;
  hbm)   timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $BENCH_SHORT
         timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $BENCH_SHORT ;;
  sq)    timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT" -- $BENCH_SHORT ;;
  pk)    JXLGPU_STREAM_PK=${PK:-1} timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT" -- $BENCH_SHORT ;;
  *) echo "usage: $0 stats|hbm|sq|pk"; exit 2 ;;
esac
find "$OUT" -name "*.csv" | head
