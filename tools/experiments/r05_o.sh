#!/bin/bash
# Round 5, GPU call O: the post stage of Modular XYB frames reading the integer planes (no float copy): parity + config 3 A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_o
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_modular.py tests/test_gpu_region.py tests/test_gpu_baseline_sizes.py -x -q 2>&1 | tail -5
for v in "" "JXLGPU_INT_POST=1" "" "JXLGPU_INT_POST=1"; do
  echo "== $v"
  env $v timeout 300 python bench.py --config 3 --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tee -a $O/cfg3_lines.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d['verified'])[:300])"
done
echo "r05_o done"
