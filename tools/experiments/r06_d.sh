#!/bin/bash
# Round 6, GPU call D: Modular suite (per-row / per-column leaves, re-squeezed residuals), ABI tests
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_d
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_modular.py tests/test_abi.py tests/test_rust_bindings.py -x -q 2>&1 | tail -25 | tee $O/modular.txt
echo "r06_d done"
