#!/bin/bash
# Round 5, GPU call H: kernel timeline (start / end per dispatch) of the overlapped batch
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_h
mkdir -p $O
cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1 NZ=0.15
FRAMES=32 REPS=2 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/tools/bench_transform.py "${1:-}" > $O/trace.log 2>&1 < /dev/null
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
out = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", "?"), short(r["Kernel_Name"])) for r in rows]
for s, e, q, n in out[-150:]:
    print(f"{s/1e3:10.1f} {e/1e3:10.1f} {(e-s)/1e3:8.1f}  q{q:>3s} {n}")
PY
rm -rf $O/tr
tail -100 $O/timeline.txt
echo "r05_h done"
