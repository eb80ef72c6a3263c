#!/bin/bash
# Round 5, GPU call K: schedule sweep at the bench's own batch size (64 frames)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_k
mkdir -p $O
cd $R
L=$R/jxl-oxide_amd/csrc
export NZ=0.15 REPS=4 FRAMES=64
V=()
for ring in 0 1; do for heavy in 0 8 12 24 28; do for side in 16 0; do
  V+=("JXLGPU_RING_MODE=$ring JXLGPU_BATCH_HEAVY=$heavy JXLGPU_TR_SIDE_MAX=$side")
done; done; done
V+=("JXLGPU_RING_MODE=0 JXLGPU_BATCH_CHUNK=32" "JXLGPU_RING_MODE=0 JXLGPU_BATCH_HEAVY=8 JXLGPU_BATCH_CHUNK=32" "JXLGPU_RING_MODE=0 JXLGPU_BATCH_HEAVY=8 JXLGPU_BATCH_CHUNK=8" "JXLGPU_NO_BATCH_OVERLAP=1" "JXLGPU_RING_MODE=0")
JXLGPU_LIB=$L/libjxlgpu.so timeout 600 python tools/bench_transform.py "${V[@]}" 2>&1 < /dev/null | grep -E "^JXLGPU|default|wall" | paste - - | sed 's/transform group.*batched all stages://' | tee $O/sweep.log
echo "r05_k done"
