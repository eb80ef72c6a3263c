#!/bin/bash
# Round 6, GPU call H: config 3 with 16-bit LDS rings in the narrow predictor kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_h
mkdir -p $O
cd $R
timeout 300 python bench.py --config 3 --frames 4 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/cfg3.json 2> $O/cfg3.err; cut -c1-700 $O/cfg3.json
timeout 600 python -m pytest tests/test_gpu_modular.py tests/test_gpu_baseline_sizes.py -x -q -k "not config2 and not config5" 2>&1 | tail -4 | tee $O/tests.txt
( cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1; rm -rf $O/st; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/bench.py --config 3 --frames 2 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify > $O/st.log 2>&1 < /dev/null )
f=$(find $O/st -name "*kernel_stats.csv" -printf '%s %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-); [ -n "$f" ] && cp "$f" $O/cfg3_kernel_stats.csv && head -12 $O/cfg3_kernel_stats.csv | cut -c1-150
rm -rf $O/st
echo "r06_h done"
