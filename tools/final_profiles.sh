set -x
# Everything behind profiles/r03_* in one gpurun call (run from the repo root on the GPU box).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python tools/first_touch.py > $O/first_touch.log 2>&1
tools/prof.sh stats > $O/stats.log 2>&1
tools/prof.sh hbm > $O/hbm.log 2>&1
tools/prof.sh sq > $O/sq.log 2>&1
tools/prof.sh clk > $O/clk.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_sq > $O/sq_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/prof_clk > $O/clk_summary.txt 2>&1
python tools/make_traffic_json.py gpurun_out/prof_hbm 8 > $O/traffic.json 2> $O/traffic.err
cd /tmp; export TMPDIR=/tmp
C3="python $R/bench.py --config 3 --frames 2 --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-extras"
C5="python $R/bench.py --config 5 --frames 4 --steps 2 --warmup 1 --no-verify --no-cpu-baseline --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3 -- $C3 > $O/cfg3_stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cfg3_fetch -- $C3 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cfg3_write -- $C3 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/cfg3_sq -- $C3 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- $C5 > $O/cfg5_stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cfg5_fetch -- $C5 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cfg5_write -- $C5 > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/cfg3_fetch $O/cfg3_write $O/cfg3_sq > $O/cfg3_pmc.txt 2>&1
python tools/pmc_summary.py $O/cfg5_fetch $O/cfg5_write > $O/cfg5_pmc.txt 2>&1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --config 3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout 400 python bench.py --config 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
ls -la $O
