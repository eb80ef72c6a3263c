#!/bin/bash
# Parity of every environment-selected kernel variant against the oracle (GPU box, repo root).
# The switches are read once at library load, hence one pytest process per variant.
#   tools/validate_variants.sh                # built-in list
#   tools/validate_variants.sh "JXLGPU_STREAM_SPLIT=2 JXLGPU_SPLIT_ROWS_A=34"
VARIANTS=("$@")
if [ ${#VARIANTS[@]} -eq 0 ]; then
  VARIANTS=("JXLGPU_STREAM_SPLIT=1" "JXLGPU_STREAM_SPLIT=2" "JXLGPU_STREAM_SPLIT=3" "JXLGPU_STREAM_PK=1" "JXLGPU_STREAM_PK=3" "JXLGPU_NO_DEQ_LUT=1" "JXLGPU_NO_STREAM=1" "JXLGPU_NO_FUSED=1")
fi
rc=0
for v in "${VARIANTS[@]}"; do
  echo "== $v"
  env $v timeout 150 python -m pytest tests/test_gpu_vardct.py tests/test_gpu_shard.py tests/test_gpu_jpeg.py tests/test_gpu_modular.py -q -m gpu -x 2>&1 | tail -2 || rc=1
done
exit $rc
