#!/bin/bash
# Round 6, GPU call I: fast sqrt / division of the HDR colour chain: device check, parity, config 5 timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_i
mkdir -p $O
cd $R
timeout 600 tools/_bin/fast_math_check 2>&1 | tee $O/fast_math_check.txt
timeout 900 python -m pytest tests/test_gpu_fast_math.py tests/test_gpu_vardct.py tests/test_gpu_baseline_sizes.py -x -q -k "not config2 and not config3" 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python bench.py --config 5 --frames 4 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/cfg5.json 2> $O/cfg5.err; cut -c1-260 $O/cfg5.json
( cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1; rm -rf $O/st; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/bench.py --config 5 --frames 2 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify > $O/st.log 2>&1 < /dev/null )
f=$(find $O/st -name "*kernel_stats.csv" -printf '%s %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-); [ -n "$f" ] && cp "$f" $O/cfg5_kernel_stats.csv && head -8 $O/cfg5_kernel_stats.csv | cut -c1-60,100-190
rm -rf $O/st
echo "r06_i done"
