#!/bin/bash
# Round 5, GPU call B: hoisted list loads, bigger items for the small shapes, one-round post launches
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_b
mkdir -p $O
cd $R
L=$R/jxl-oxide_amd/csrc
export NZ=0.15 FRAMES=32 REPS=6
par() { # lib: parity of the list-fed kernels under that library
  JXLGPU_LIB=$L/$1 timeout 300 python -m pytest tests/test_gpu_grouped.py tests/test_gpu_batch.py tests/test_gpu_vardct.py -x -q -p no:cacheprovider 2>&1 < /dev/null | tail -2
}
run() { # lib, variants...
  lib=$1; shift
  echo "=== lib $lib" | tee -a $O/sweep.log
  JXLGPU_LIB=$L/$lib timeout 300 python tools/bench_transform.py "$@" 2>&1 < /dev/null | grep -v "^CANARY" | tee -a $O/sweep.log
}
par libjxlgpu.so; par libjxlgpu_nbi2.so; par libjxlgpu_nbi2a.so
run libjxlgpu.so "" "JXLGPU_NO_BATCH_OVERLAP=1" "JXLGPU_BATCH_STREAM_ROWS=96" "JXLGPU_BATCH_HEAVY=24" "JXLGPU_BATCH_CHUNK=32" "JXLGPU_BATCH_CHUNK=8"
run libjxlgpu_nbi2.so "" "JXLGPU_NO_BATCH_OVERLAP=1"
run libjxlgpu_nbi2a.so "" "JXLGPU_NO_BATCH_OVERLAP=1"
run libjxlgpu_np0.so "" "JXLGPU_NO_BATCH_OVERLAP=1"
run libjxlgpu_np2.so "" "JXLGPU_NO_BATCH_OVERLAP=1"
run libjxlgpu.so "" "JXLGPU_NO_BATCH_OVERLAP=1"
echo "r05_b done"
