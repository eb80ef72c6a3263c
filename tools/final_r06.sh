#!/bin/bash
# Round-6 records (run from the repo root on the GPU box through gpurun; every step bounded).  usage: tools/final_r06.sh A|B|C
#   A: the driver's bench command + rocprofv3 kernel stats / SQ counters / HBM counters (with the mem_probe calibration passes) of the same command line (short form)
#   B: configs 3 and 5 (bench lines + kernel stats)
#   C: the whole -m gpu suite + the N = 2 bench path on one device
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_r06
mkdir -p $O
cd $R
B="python $R/bench.py --steps 3 --warmup 1 --distinct 2 --no-cpu-baseline --no-extras --no-verify"
biggest() { find "$1" -name "$2" -printf '%s %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-; }
prof() { # mode, outdir, command...
  mode=$1; out=$2; shift 2
  rm -rf "$out"; mkdir -p "$out"
  ( cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1
    case "$mode" in
      stats) timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- "$@" > "$out/cmd.log" 2>&1 < /dev/null ;;
      sq)    timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d "$out" -- "$@" > "$out/cmd.log" 2>&1 < /dev/null ;;
      fetch) timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out" -- "$@" > "$out/cmd.log" 2>&1 < /dev/null ;;
      write) timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out" -- "$@" > "$out/cmd.log" 2>&1 < /dev/null ;;
    esac )
}
case "$1" in
  A) timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null; echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
     prof stats $O/stats $B; f=$(biggest $O/stats "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -12 $O/bench_kernel_stats.csv | cut -c1-150
     prof sq $O/sq $B --frames 8; python tools/pmc_summary.py $O/sq > $O/pmc_sq_counters.txt 2>&1 < /dev/null; cut -c1-200 $O/pmc_sq_counters.txt
     mkdir -p $O/hbm; prof fetch $O/hbm/fetch $B --frames 8; prof write $O/hbm/write $B --frames 8
     # the FETCH x 2 calibration, re-taken this round: tools/mem_probe.hip moves a known byte count per kernel
     /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/mem_probe.hip -o /tmp/mem_probe 2>/dev/null
     if [ -x /tmp/mem_probe ]; then
       timeout 120 /tmp/mem_probe > $O/mem_probe.txt 2>&1; tail -12 $O/mem_probe.txt
       prof fetch $O/hbm/probe_fetch /tmp/mem_probe; prof write $O/hbm/probe_write /tmp/mem_probe
     fi
     python tools/make_traffic_json.py $O/hbm 8 > $O/pmc_hbm_traffic.json 2>$O/traffic.err; head -c 1200 $O/pmc_hbm_traffic.json
     rm -rf $O/stats $O/sq/*/ 2>/dev/null
     find $O -name "*.csv" -size +8M -delete ;;
  B) for c in 3 5; do
       timeout 900 python bench.py --config $c --cpu-seconds 2 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err < /dev/null; echo "cfg$c rc=$?"; cut -c1-300 $O/bench_cfg$c.json
       prof stats $O/stats_cfg$c python $R/bench.py --config $c --frames 2 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify
       f=$(biggest $O/stats_cfg$c "*kernel_stats.csv"); [ -n "$f" ] && cp "$f" $O/cfg${c}_kernel_stats.csv && head -14 $O/cfg${c}_kernel_stats.csv | cut -c1-150
       rm -rf $O/stats_cfg$c
     done ;;
  C) timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_suite.txt
     JXLGPU_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n2_one_device.json 2> $O/bench_n2.err < /dev/null
     echo "n2 rc=$?"; cut -c1-400 $O/bench_n2_one_device.json; tail -3 $O/bench_n2.err ;;
esac
echo "final_r06 $1 done"
