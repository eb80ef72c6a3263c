#!/bin/bash
# Round 6, GPU call E: kernel timeline (start / end per dispatch) of the overlapped batch with the current build; extended VALU probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_e
mkdir -p $O
cd $R && timeout 200 tools/_bin/valu_cost_probe > $O/valu_cost_probe.txt 2>&1; tail -45 $O/valu_cost_probe.txt
cd /tmp && export TMPDIR=/tmp JXLGPU_NO_CANARY=1 NZ=0.15
FRAMES=64 REPS=2 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/tools/bench_transform.py "${1:-}" > $O/trace.log 2>&1 < /dev/null
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 - "$f" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
out = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", "?"), short(r["Kernel_Name"])) for r in rows]
for s, e, q, n in out[-130:]:
    print(f"{s/1e3:10.1f} {e/1e3:10.1f} {(e-s)/1e3:8.1f}  q{q:>3s} {n}")
PY
rm -rf $O/tr
tail -90 $O/timeline.txt
echo "r06_e done"
