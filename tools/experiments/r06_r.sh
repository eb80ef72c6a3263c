#!/bin/bash
# Round 6, GPU call R: kernel timeline of config 3 (one 8K Modular frame)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py --config 3 --frames 2 --distinct 1 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify > $O/trace.log 2>&1 < /dev/null
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
out = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", "?"), short(r["Kernel_Name"])) for r in rows]
for s, e, q, n in out[-75:]:
    print(f"{s/1e3:10.1f} {e/1e3:10.1f} {(e-s)/1e3:8.1f}  q{q:>3s} {n}")
PY
rm -rf $O/tr
cat $O/timeline.txt
echo "r06_r done"
