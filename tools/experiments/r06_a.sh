#!/bin/bash
# Round 6, GPU call A: instruction-cost probe, packed probe (re-taken), and three schedule / occupancy experiments on the batched job
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_a
mkdir -p $O
cd $R
timeout 120 tools/_bin/valu_cost_probe 2>&1 | tee $O/valu_cost_probe.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/pk_probe.hip -o /tmp/pk_probe 2>/dev/null && timeout 120 /tmp/pk_probe 2>&1 | tee $O/pk_probe.txt
FRAMES=64 NZ=0.15 REPS=6 timeout 900 python tools/bench_transform.py "" "JXLGPU_BATCH_LF_AHEAD=1" "JXLGPU_POST_LDS_PAD=81920" "JXLGPU_POST_LDS_PAD=81920 JXLGPU_NO_BATCH_OVERLAP=1" "JXLGPU_NO_BATCH_OVERLAP=1" "" 2>&1 | tee $O/sched.txt
echo "r06_a done"
