#!/bin/bash
# Round 6, GPU call Y: the D = 4 predictor step after its second pass (row starts at every fourth step only, 24-bit multiply-adds,
# branch-free clamp, range guard once per group), default scheduler against -amdgpu-sched-strategy=max-ilp (libjxlgpu_sched.so),
# parity first.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_y
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_modular.py tests/test_gpu_baseline_sizes.py -m gpu -q -x < /dev/null 2>&1 | tail -3 | tee $O/tests.txt
JXLGPU_LIB=$R/jxl-oxide_amd/csrc/libjxlgpu_sched.so timeout 600 python -m pytest tests/test_gpu_modular.py -m gpu -q -x < /dev/null 2>&1 | tail -3 | tee $O/tests_sched.txt
for lib in libjxlgpu.so libjxlgpu_sched.so libjxlgpu.so libjxlgpu_sched.so; do
  JXLGPU_LIB=$R/jxl-oxide_amd/csrc/$lib timeout 400 python bench.py --config 3 --frames 8 --distinct 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras < /dev/null > $O/cfg3_$lib.json 2> $O/cfg3.err
  echo "$lib: $(python -c "import json,sys; d=json.loads([l for l in open('$O/cfg3_$lib.json') if l.startswith('{')][-1]); print(d['value'], d['unit'], d['ms_per_step'], (d.get('verified') or {}).get('ok'))")"
done
cd /tmp && export TMPDIR=/tmp
for lib in libjxlgpu.so libjxlgpu_sched.so; do
  JXLGPU_LIB=$R/jxl-oxide_amd/csrc/$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$lib -o p -- python $R/bench.py --config 3 --frames 8 --distinct 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-verify < /dev/null > $O/prof_$lib.log 2>&1
  f=$(find $O/prof_$lib -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    echo "-- $lib"; grep predict_ "$f" | sed 's/(anonymous namespace):://g' | cut -c1-60,120-200
    cp "$f" $O/kernel_stats_$lib.csv
  fi
  rm -rf $O/prof_$lib
done
echo "r06_y done"
