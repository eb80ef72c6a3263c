#!/bin/bash
# Round 6, GPU call Q: interior fast paths of the LDS tile kernel (no mirror / edge tests where the halo lies inside the image): parity + config 5
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_q
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_vardct.py tests/test_gpu_region.py tests/test_gpu_modular.py tests/test_gpu_baseline_sizes.py tests/test_gpu_schedules.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
for i in 1 2; do timeout 300 python bench.py --config 5 --frames 8 --distinct 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/cfg5.json 2> $O/cfg5.err; echo "cfg5: $(cut -c95-200 $O/cfg5.json)"; done
( cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1; rm -rf $O/st; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/bench.py --config 5 --frames 2 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify > $O/st.log 2>&1 < /dev/null )
f=$(find $O/st -name "*kernel_stats.csv" -printf '%s %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-); [ -n "$f" ] && cp "$f" $O/cfg5_kernel_stats.csv && head -5 $O/cfg5_kernel_stats.csv | cut -c1-60,100-190
rm -rf $O/st
echo "r06_q done"
