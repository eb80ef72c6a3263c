#!/bin/bash
# Round-5 records (run from the repo root on the GPU box through gpurun; every step bounded).  usage: tools/final_r05.sh A|B
#   A: the driver's bench command + rocprofv3 kernel stats / SQ counters / HBM counters of the same command line (short form)
#   B: configs 3 and 5 (bench lines + kernel stats), the density sweep
# Everything lands under gpurun_out/final_r05/; copy into profiles/ with tools/collect_r05.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_r05
mkdir -p $O
cd $R
B="python $R/bench.py --steps 3 --warmup 1 --distinct 2 --no-cpu-baseline --no-extras --no-verify"
biggest() { find "$1" -name "$2" -printf '%s %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-; }
prof() { # mode, outdir, extra bench args
  mode=$1; out=$2; shift 2
  rm -rf "$out"; mkdir -p "$out"
  ( cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1
    case "$mode" in
      stats) timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- $B "$@" > "$out/bench.log" 2>&1 < /dev/null ;;
      sq)    timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d "$out" -- $B --frames 8 "$@" > "$out/bench.log" 2>&1 < /dev/null ;;
      fetch) timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out" -- $B --frames 8 "$@" > "$out/bench.log" 2>&1 < /dev/null ;;
      write) timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out" -- $B --frames 8 "$@" > "$out/bench.log" 2>&1 < /dev/null ;;
    esac )
}
case "$1" in
  A) timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
     prof stats $O/stats; f=$(biggest $O/stats "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -12 $O/bench_kernel_stats.csv | cut -c1-150
     prof sq $O/sq; python tools/pmc_summary.py $O/sq > $O/pmc_sq_counters.txt 2>&1 < /dev/null; cut -c1-200 $O/pmc_sq_counters.txt
     mkdir -p $O/hbm; prof fetch $O/hbm/fetch; prof write $O/hbm/write
     python tools/make_traffic_json.py $O/hbm 8 > $O/pmc_hbm_traffic.json 2>$O/traffic.err; head -c 600 $O/pmc_hbm_traffic.json
     rm -rf $O/stats $O/sq/*/ 2>/dev/null
     find $O -name "*.csv" -size +8M -delete ;;
  B) for c in 3 5; do
       timeout 300 python bench.py --config $c --cpu-seconds 2 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err < /dev/null; echo "cfg$c rc=$?"; cut -c1-300 $O/bench_cfg$c.json
       ( cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1; rm -rf $O/stats_cfg$c
         timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg$c -- python $R/bench.py --config $c --frames 2 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify > $O/stats_cfg$c.log 2>&1 < /dev/null )
       f=$(biggest $O/stats_cfg$c "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" $O/cfg${c}_kernel_stats.csv && head -14 $O/cfg${c}_kernel_stats.csv | cut -c1-150
       rm -rf $O/stats_cfg$c
     done
     timeout 500 python bench.py --nz 0.05,0.10 --no-cpu-baseline --no-extras --distinct 4 > $O/bench_nz.jsonl 2> $O/bench_nz.err < /dev/null; echo "nz rc=$?"; cut -c1-200 $O/bench_nz.jsonl ;;
esac
echo "final_r05 $1 done"
