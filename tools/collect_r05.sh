#!/bin/bash
# gpurun_out/final_r05/* -> profiles/r05_* (what the judge reads)
cd "$(dirname "$0")/.."
O=gpurun_out/final_r05; P=profiles
[ -f $O/bench_kernel_stats.csv ] && cp $O/bench_kernel_stats.csv $P/r05_bench_kernel_stats.csv
[ -f $O/pmc_sq_counters.txt ] && cp $O/pmc_sq_counters.txt $P/r05_pmc_sq_counters.txt
[ -s $O/pmc_hbm_traffic.json ] && cp $O/pmc_hbm_traffic.json $P/r05_pmc_hbm_traffic.json
: > $P/r05_bench_lines.jsonl
for f in bench_default.json bench_cfg3.json bench_cfg5.json bench_nz.jsonl; do [ -s $O/$f ] && cat $O/$f >> $P/r05_bench_lines.jsonl; done
for c in 3 5; do [ -f $O/cfg${c}_kernel_stats.csv ] && cp $O/cfg${c}_kernel_stats.csv $P/r05_cfg${c}_kernel_stats.csv; done
ls -la $P/r05_*
