#!/bin/bash
# Round-4 records in bounded steps (run from the repo root on the GPU box).  usage: tools/final_r04.sh A|B|C
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_r04
mkdir -p $O
cd $R
case "$1" in
  A) # correctness: the whole suite, then the fault hunt (fresh-process loops + the suite under both guard modes)
     timeout 400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
     rm -f gpurun_out/hunt/hunt.log
     timeout 900 python tools/fault_hunt.py --abi 120 --smoke 30 --suite > $O/fault_hunt.out 2>&1 < /dev/null; echo "hunt rc=$?"
     cp gpurun_out/hunt/hunt.log $O/fault_hunt.log; tail -40 $O/fault_hunt.log | cut -c1-180 ;;
  B) # the headline bench as the driver runs it, then the profiles of the same command line (short form)
     timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err < /dev/null; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
     tools/prof_r04.sh stats > $O/stats.log 2>&1; cp gpurun_out/prof_r04_stats/kernel_stats.csv $O/ 2>/dev/null; head -12 $O/kernel_stats.csv | cut -c1-150
     tools/prof_r04.sh sq > $O/sq.log 2>&1; python tools/pmc_summary.py gpurun_out/prof_r04_sq > $O/sq_summary.txt 2>&1 < /dev/null
     tools/prof_r04.sh hbm > $O/hbm.log 2>&1
     mkdir -p $O/hbm/fetch $O/hbm/write; cp gpurun_out/prof_r04_hbm/FETCH_SIZE.csv $O/hbm/fetch/x_counter_collection.csv; cp gpurun_out/prof_r04_hbm/WRITE_SIZE.csv $O/hbm/write/x_counter_collection.csv
     ls -la $O $O/hbm/* | head -30 ;;
  C) # the other configurations and the density sweep
     timeout 300 python bench.py --config 3 --cpu-seconds 2 > $O/bench_cfg3.json 2> $O/bench_cfg3.err < /dev/null; echo "cfg3 rc=$?"; cut -c1-300 $O/bench_cfg3.json
     timeout 300 python bench.py --config 5 --cpu-seconds 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err < /dev/null; echo "cfg5 rc=$?"; cut -c1-300 $O/bench_cfg5.json
     timeout 500 python bench.py --nz 0.05,0.10 --no-cpu-baseline --distinct 4 > $O/bench_nz.jsonl 2> $O/bench_nz.err < /dev/null; echo "nz rc=$?"; cut -c1-200 $O/bench_nz.jsonl ;;
esac
echo "final $1 done"
