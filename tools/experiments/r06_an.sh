#!/bin/bash
# Round 6, last records on the final tree: the -m gpu suite, smoke, the config-3 line + kernel stats, the driver's bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_an
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q < /dev/null 2>&1 | tail -2 | tee $O/gpu_suite.txt
python __graft_entry__.py smoke 2>&1 | tail -1 | tee -a $O/gpu_suite.txt
timeout 600 python bench.py --config 3 --cpu-seconds 2 < /dev/null > $O/bench_cfg3.json 2> $O/bench_cfg3.err; cut -c1-200 $O/bench_cfg3.json
timeout 900 python bench.py < /dev/null > $O/bench_default.json 2> $O/bench_default.err; cut -c1-200 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1
JXLGPU_BENCH_CONTEXTS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o p -- python $R/bench.py --config 3 --frames 8 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify < /dev/null > $O/st.log 2>&1
f=$(find $O/st -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cfg3_kernel_stats_one_context.csv && head -4 "$f" | cut -c1-40,150-230
rm -rf $O/st
echo "r06_an done"
