#!/bin/bash
# Round 6, GPU call P: config 5 with the two output rows of the upsampling kernel's colour epilogue unrolled (libjxlgpu_ym2.so) against the product
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_p
mkdir -p $O
cd $R
for lib in "" ym2 "" ym2; do
  if [ -n "$lib" ]; then export JXLGPU_LIB=$R/jxl-oxide_amd/csrc/libjxlgpu_$lib.so; else unset JXLGPU_LIB; fi
  timeout 300 python bench.py --config 5 --frames 8 --distinct 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/cfg5_$lib.json 2> $O/cfg5.err; echo "lib=${lib:-product}: $(cut -c95-200 $O/cfg5_$lib.json)"
done
echo "r06_p done"
