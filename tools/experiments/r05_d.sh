#!/bin/bash
# Round 5, GPU call D: post kernel at 168 VGPRs (filter constants in SGPR pairs, three waves per SIMD) against the 193-register form
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_d
mkdir -p $O
cd $R
L=$R/jxl-oxide_amd/csrc
export NZ=0.15 FRAMES=32 REPS=6
par() { JXLGPU_LIB=$L/$1 timeout 300 python -m pytest tests/test_gpu_batch.py tests/test_gpu_vardct.py tests/test_gpu_baseline_sizes.py -x -q -p no:cacheprovider -k "not modular and not squeeze and not predictor" 2>&1 < /dev/null | tail -2; }
run() { lib=$1; shift; echo "=== lib $lib" | tee -a $O/sweep.log; JXLGPU_LIB=$L/$lib timeout 300 python tools/bench_transform.py "$@" 2>&1 < /dev/null | grep -v "^CANARY" | tee -a $O/sweep.log; }
par libjxlgpu.so
run libjxlgpu.so "" "JXLGPU_NO_BATCH_OVERLAP=1" "JXLGPU_BATCH_STREAM_ROWS=536" "JXLGPU_BATCH_STREAM_ROWS=268" "JXLGPU_BATCH_STREAM_ROWS=180" "JXLGPU_BATCH_STREAM_ROWS=716" "JXLGPU_BATCH_HEAVY=24" "JXLGPU_BATCH_CHUNK=8" "JXLGPU_BATCH_CHUNK=24" "JXLGPU_BATCH_CHUNK=32"
run libjxlgpu_pk193.so "" "JXLGPU_NO_BATCH_OVERLAP=1"
echo "r05_d done"
