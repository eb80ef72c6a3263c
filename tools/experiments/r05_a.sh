#!/bin/bash
# Round 5, GPU call A: parity of the changed transform kernels, then the schedule / kernel-variant sweep
# (tools/bench_transform.py: 32 frames of one 4K workload at 15 % non-zeros, batched all stages, wall clock per frame)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_a
mkdir -p $O
cd $R
export JXLGPU_NO_CANARY=
L=$R/jxl-oxide_amd/csrc
timeout 300 python -m pytest tests/test_gpu_canary.py tests/test_gpu_grouped.py tests/test_gpu_batch.py tests/test_gpu_vardct.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $O/pytest.log
export NZ=0.15 FRAMES=32 REPS=6
run() { # lib, variants...
  lib=$1; shift
  echo "=== lib $lib" | tee -a $O/sweep.log
  JXLGPU_LIB=$L/$lib timeout 300 python tools/bench_transform.py "$@" 2>&1 < /dev/null | grep -v "^CANARY" | tee -a $O/sweep.log
}
run libjxlgpu.so "" "JXLGPU_NO_BATCH_OVERLAP=1" "JXLGPU_BATCH_HEAVY=24" "JXLGPU_BATCH_HEAVY=28" "JXLGPU_BATCH_HEAVY=8" "JXLGPU_BATCH_HEAVY=16" \
    "JXLGPU_BATCH_HEAVY=24 JXLGPU_BATCH_STREAM_ROWS=536" "JXLGPU_BATCH_STREAM_ROWS=536" "JXLGPU_BATCH_HEAVY=24 JXLGPU_BATCH_CHUNK=8" "JXLGPU_BATCH_HEAVY=24 JXLGPU_BATCH_CHUNK=32" "JXLGPU_BATCH_HEAVY=31"
run libjxlgpu_old.so "" "JXLGPU_NO_BATCH_OVERLAP=1" "JXLGPU_BATCH_HEAVY=24"
run libjxlgpu_su.so "" "JXLGPU_NO_BATCH_OVERLAP=1"
run libjxlgpu_np2.so "" "JXLGPU_NO_BATCH_OVERLAP=1"
echo "=== phase stamps (prof lib, one stream)" | tee -a $O/sweep.log
FRAMES=16 REPS=2 JXLGPU_LIB=$L/libjxlgpu_prof.so timeout 200 python tools/bench_transform.py "JXLGPU_NO_BATCH_OVERLAP=1" 2>&1 < /dev/null | grep -E "tr_prof|wall" | tee -a $O/sweep.log
echo "r05_a done"
