#!/bin/bash
# Round 5, final records of the second session: the whole -m gpu suite, the N = 2 bench path on one device (both ranks on GPU 0:
# sharding, peer-write gather, verification of a gathered frame), the driver's bench command, rocprofv3 kernel stats of the same command
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_q; mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/suite.txt
JXLGPU_BENCH_ONE_DEVICE=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
   bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_n2_one_device.json 2> $O/bench_n2.err < /dev/null
echo "n2 rc=$?"; cut -c1-400 $O/bench_n2_one_device.json; tail -3 $O/bench_n2.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err < /dev/null
echo "bench rc=$?"; cut -c1-300 $O/bench_default.json
B="python $R/bench.py --steps 3 --warmup 1 --distinct 2 --no-cpu-baseline --no-extras --no-verify"
( cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1; rm -rf $O/stats
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1 < /dev/null )
f=$(find $O/stats -name "*kernel_stats.csv" -printf '%s %p\n' 2>/dev/null | sort -n | tail -1 | cut -d' ' -f2-)
[ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -8 $O/bench_kernel_stats.csv | cut -c1-160
rm -rf $O/stats
echo "r05_q done"
