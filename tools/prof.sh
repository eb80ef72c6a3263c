#!/bin/bash
# The rocprofv3 invocations behind profiles/r02_* / r03_* (run on the GPU box from the repo root, e.g. through
# gpurun).  Counters are collected in their own passes with --kernel-trace only.
#   tools/prof.sh stats   -> per-kernel durations of the headline bench (the driver's command)
#   tools/prof.sh hbm     -> FETCH_SIZE / WRITE_SIZE (two passes) of the bench (8 frames) AND of
#                            tools/_mem_probe (known byte counts: calibration of both counters)
#   tools/prof.sh sq      -> SQ issue / wait counters of the bench kernels
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_SHORT="python $R/bench.py --steps 2 --warmup 1 --frames 8 --no-cpu-baseline --no-extras --no-verify --distinct 2"
case "$1" in
  stats) timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras ;;
  hbm)   timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $BENCH_SHORT
         timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $BENCH_SHORT
         timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/probe_fetch" -- $R/tools/_mem_probe
         timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/probe_write" -- $R/tools/_mem_probe ;;
  sq)    timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT" -- $BENCH_SHORT ;;
  clk)   # GRBM_GUI_ACTIVE = cycles the GPU was busy during the dispatch: / duration = the clock the kernel really ran at
         timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d "$OUT" -- $BENCH_SHORT ;;
  *) echo "usage: $0 stats|hbm|sq|clk"; exit 2 ;;
esac
find "$OUT" -name "*.csv" | head -3
