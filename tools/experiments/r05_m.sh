#!/bin/bash
# Round 5, GPU call M: predictor waves of the top Squeeze levels beside the deep levels (config 3); host threads of the upload
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_m
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_modular.py tests/test_gpu_baseline_sizes.py -x -q -p no:cacheprovider 2>&1 < /dev/null | tail -2
for v in "JXLGPU_PRED_LATE_STEPS=0" "JXLGPU_PRED_LATE_STEPS=1" "JXLGPU_PRED_LATE_STEPS=2" "JXLGPU_PRED_LATE_STEPS=3" "JXLGPU_PRED_LATE_STEPS=0" "JXLGPU_PRED_LATE_STEPS=1"; do
  echo "== $v"
  env $v timeout 300 python bench.py --config 3 --frames 4 --distinct 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"verified\"][\"ok\"], d[\"roofline\"][\"group_ms_per_frame\"])"
done
for t in 3 7 15 31; do
  echo "== JXLGPU_HOST_THREADS=$t"
  JXLGPU_HOST_THREADS=$t NZ=0.15 timeout 200 python tools/e2e_split.py 2>&1 | tail -4
done
echo "r05_m done"
