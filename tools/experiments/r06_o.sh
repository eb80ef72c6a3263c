#!/bin/bash
# Round 6, GPU call O: phase stamps of the transform families (make PROF=1: s_memtime per phase, summed per family), alone and under the overlap
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_o
mkdir -p $O
cd $R
export JXLGPU_LIB=$R/jxl-oxide_amd/csrc/libjxlgpu_prof.so
echo "== stages one after the other (JXLGPU_NO_BATCH_OVERLAP=1)" | tee $O/phases.txt
FRAMES=32 NZ=0.15 REPS=3 timeout 300 python tools/bench_transform.py "JXLGPU_NO_BATCH_OVERLAP=1" 2>&1 | grep -v "^CANARY" | tee -a $O/phases.txt
echo "== default schedule (transform launches of chunk k + 1 beside the post launch of chunk k)" | tee -a $O/phases.txt
FRAMES=64 NZ=0.15 REPS=3 timeout 300 python tools/bench_transform.py "" 2>&1 | grep -v "^CANARY" | tee -a $O/phases.txt
echo "r06_o done"
