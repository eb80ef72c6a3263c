#!/bin/bash
# Round 6, GPU call AM: the predictor kernel storing 16 samples per lane at a time: parity, config 3 (one / three contexts), WRITE_SIZE again
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_am
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_modular.py tests/test_gpu_baseline_sizes.py tests/test_gpu_schedules.py -m gpu -q -x < /dev/null 2>&1 | tail -3 | tee $O/tests.txt
one() { # tag, env
  env $2 timeout 400 python bench.py --config 3 --frames 12 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/$1.json 2> $O/err.txt
  echo "$1 [$2]: $(python -c "import json,sys; d=json.loads([l for l in open('$O/$1.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"
}
one c3 JXLGPU_BENCH_CONTEXTS=3
one c1 JXLGPU_BENCH_CONTEXTS=1
one c3b JXLGPU_BENCH_CONTEXTS=3
one c5 JXLGPU_BENCH_CONTEXTS=5
timeout 200 python tests/tools/fuzz_parity.py 60 6401 < /dev/null 2>&1 | tail -2 | cut -c1-300
bash tools/experiments/r06_al.sh 2>&1 | grep -E "predict_lanes_wp4|config 3" | cut -c1-200
cp gpurun_out/r06_al/summary.txt $O/pmc_summary.txt
echo "r06_am done"
