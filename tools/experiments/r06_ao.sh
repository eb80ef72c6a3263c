#!/bin/bash
# Round 6, GPU call AO: vertical Squeeze steps of few waves with one column per lane
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
one() { env $1 timeout 300 python bench.py --config 3 --frames 12 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['verified']['ok'])"; }
one "JXLGPU_BENCH_CONTEXTS=1 JXLGPU_SQZ_V2_MIN_WAVES=0"
one "JXLGPU_BENCH_CONTEXTS=1 JXLGPU_SQZ_V2_MIN_WAVES=2048"
one "JXLGPU_BENCH_CONTEXTS=1 JXLGPU_SQZ_V2_MIN_WAVES=8192"
one "JXLGPU_BENCH_CONTEXTS=1 JXLGPU_SQZ_V2_MIN_WAVES=1000000"
one "JXLGPU_BENCH_CONTEXTS=3 JXLGPU_SQZ_V2_MIN_WAVES=0"
one "JXLGPU_BENCH_CONTEXTS=3 JXLGPU_SQZ_V2_MIN_WAVES=8192"
