#!/usr/bin/env python3
"""Mean counter value per dispatch per kernel from rocprofv3 --pmc counter_collection.csv files.
    python tools/pmc_summary.py gpurun_out/pmc_a [more dirs...]"""
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (did, k), cs in per.items():
            for c, v in cs.items():
                acc[k][c].append(v)
for k, cs in acc.items():
    if "rocclr" in k or "at::native" in k:
        continue
    name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    print(name[:60], " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())), f"n={len(next(iter(cs.values())))}")
