#!/bin/bash
# Round 6, GPU call N: issue priority by chain length in the narrow predictor kernel (config 3)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_n
mkdir -p $O
cd $R
for v in 1 0 1 0; do
  JXLGPU_PRED_PRIO=$v timeout 300 python bench.py --config 3 --frames 4 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/cfg3_prio$v.json 2> $O/cfg3.err; echo "PRED_PRIO=$v: $(cut -c1-150 $O/cfg3_prio$v.json)"
done
timeout 600 python -m pytest tests/test_gpu_modular.py tests/test_gpu_baseline_sizes.py -x -q -k "not config2 and not config5" 2>&1 | tail -3 | tee $O/tests.txt
( cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1; rm -rf $O/st; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python $R/bench.py --config 3 --frames 2 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify > $O/st.log 2>&1 < /dev/null )
f=$(find $O/st -name "*kernel_stats.csv" -printf '%s %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-); [ -n "$f" ] && cp "$f" $O/cfg3_kernel_stats.csv && head -4 $O/cfg3_kernel_stats.csv | cut -c1-60,150-260
rm -rf $O/st
echo "r06_n done"
