#!/bin/bash
# Round 5, GPU call L: the non-bit-exact post option (speed and ULP histogram), and the default line with other_configs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_l
mkdir -p $O
cd $R
for v in "" "JXLGPU_POST_FAST=1" ""; do
  echo "== $v"
  env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 15 2>/dev/null | tee -a $O/lines.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"frac\"], d[\"roofline\"][\"group_ms_per_frame\"], json.dumps(d[\"verified\"])[:600], d[\"roofline_valu\"])"
done
timeout 900 python bench.py 2>$O/default.err | tee $O/default.json | cut -c1-300
python -c "
import json
d=json.loads(open('$O/default.json').read())
print(json.dumps(d['other_configs'])[:1500]); print(json.dumps(d['end_to_end'])[:600]); print(json.dumps(d['cpu_baseline'])[:400])"
echo "r05_l done"
