#!/bin/bash
# Round 6, GPU call AA: the waves of config 3's predictor pass (how many chains of which length share the 1024 SIMDs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_aa
mkdir -p $O
cd $R
JXLGPU_PRED_DUMP=1 timeout 300 python bench.py --config 3 --frames 1 --distinct 1 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-verify < /dev/null > $O/out.json 2> $O/err.txt
grep predwave $O/err.txt | sort -u -k2,2n -t' ' | awk '{print $3, $4, $5, $6, $7, $8}' | sort | uniq -c | sort -k7 -t= -rn > $O/waves.txt
wc -l $O/waves.txt; head -50 $O/waves.txt
