set -x
# Everything behind profiles/r02_* in one gpurun call (run from the repo root on the GPU box).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/final
cd $R
python tools/first_touch.py > gpurun_out/final/first_touch.log 2>&1
tools/prof.sh stats > gpurun_out/final/stats.log 2>&1
tools/prof.sh hbm > gpurun_out/final/hbm.log 2>&1
tools/prof.sh sq > gpurun_out/final/sq.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_sq > gpurun_out/final/sq_summary.txt 2>&1
python tools/make_traffic_json.py gpurun_out/prof_hbm 8 > gpurun_out/final/traffic.json 2> gpurun_out/final/traffic.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_mod -- python $R/tools/bench_modular.py > $R/gpurun_out/final/bench_modular.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/mod_fetch -- python $R/tools/bench_modular.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/mod_write -- python $R/tools/bench_modular.py > /dev/null 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/final/mod_fetch gpurun_out/final/mod_write > gpurun_out/final/mod_traffic.txt 2>&1
timeout 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
timeout 400 python bench.py --config 3 > gpurun_out/final/bench_cfg3.json 2> gpurun_out/final/bench_cfg3.err
timeout 400 python bench.py --config 5 > gpurun_out/final/bench_cfg5.json 2> gpurun_out/final/bench_cfg5.err
ls -la gpurun_out/final
