#!/bin/bash
# Round 6, GPU call AJ: per-launch durations of the inverse Squeeze of one 8K frame (kernel trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_aj
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
JXLGPU_BENCH_CONTEXTS=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python $R/bench.py --config 3 --frames 1 --distinct 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify < /dev/null > $O/log.txt 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame: the dispatches after the last post_pk_kernel but one
idx = [i for i, r in enumerate(rows) if "post_pk_kernel" in r["Kernel_Name"]]
lo = idx[-2] + 1 if len(idx) >= 2 else 0
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:idx[-1] + 1]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-46:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s}x{r.get('Grid_Size_Y', ''):>5s}x{r.get('Grid_Size_Z', ''):>2s}  {n}")
PY
wc -l $O/timeline.txt; rm -rf $O/tr
