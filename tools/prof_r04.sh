#!/bin/bash
# Round-4 profiling passes (run on the GPU box from the repo root through gpurun); every command is bounded and
# reads nothing from stdin.  usage: tools/prof_r04.sh stats|sq|hbm [extra bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
MODE=$1; shift
OUT=$R/gpurun_out/prof_r04_$MODE
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1
B="python $R/bench.py --steps 3 --warmup 1 --distinct 2 --no-cpu-baseline --no-extras --no-verify $*"
biggest() { find "$1" -name "$2" -printf '%s %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-; }
case "$MODE" in
  stats) timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- $B > "$OUT/bench.log" 2>&1 < /dev/null
         f=$(biggest "$OUT" "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -16 "$f" | cut -c1-170 ;;
  sq)    timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT" -- $B --frames 8 > "$OUT/bench.log" 2>&1 < /dev/null
         f=$(biggest "$OUT" "*counter_collection.csv"); [ -n "$f" ] && cp "$f" "$OUT/counters.csv" && python $R/tools/pmc_summary.py "$OUT" 2>&1 < /dev/null | head -40 | cut -c1-220 ;;
  hbm)   for c in FETCH_SIZE WRITE_SIZE; do
           timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -- $B --frames 8 > "$OUT/bench_$c.log" 2>&1 < /dev/null
           f=$(biggest "$OUT/$c" "*counter_collection.csv"); [ -n "$f" ] && cp "$f" "$OUT/$c.csv"
         done
         ls -la "$OUT" ;;
esac
tail -2 "$OUT"/bench*.log 2>/dev/null | cut -c1-200
echo "prof $MODE done"
