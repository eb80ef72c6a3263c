#!/usr/bin/env python3
"""profiles/r02_pmc_hbm_traffic.json from the `tools/prof.sh hbm` passes: HBM-side bytes per frame
of every bench kernel (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, both in KB),
with the corrections calibrated in the same session on tools/_mem_probe (known byte counts):
FETCH_SIZE reports half of the bytes read for 16-byte AND 4-byte per-lane loads (x2), WRITE_SIZE
is exact (x1).

    python tools/make_traffic_json.py gpurun_out/prof_hbm 8 > profiles/r02_pmc_hbm_traffic.json"""
import collections, csv, glob, json, os, sys


def mean_per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for did, v in per.items():
            n = names[did].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            acc[n].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    root, frames = sys.argv[1], int(sys.argv[2])
    fetch = mean_per_kernel(os.path.join(root, "fetch"), "FETCH_SIZE")
    write = mean_per_kernel(os.path.join(root, "write"), "WRITE_SIZE")
    pf = mean_per_kernel(os.path.join(root, "probe_fetch"), "FETCH_SIZE")   # optional: the calibration passes
    pw = mean_per_kernel(os.path.join(root, "probe_write"), "WRITE_SIZE")
    plane_mb = 3840 * 2176 * 12 / 1e6
    calib = {k: {"bytes_read_MB": round(plane_mb, 2), "FETCH_SIZE_MB": round(pf[k] * 1024 / 1e6, 2),
                 "bytes_written_MB": round(plane_mb, 2), "WRITE_SIZE_MB": round(pw[k] * 1024 / 1e6, 2)}
             for k in ("copy16", "walk_rowmajor", "walk_tiled") if k in pf and k in pw}
    if not calib:
        calib = "not repeated in this session: see profiles/r03_pmc_hbm_traffic.json (same image, same counters)"
    out = {"units": "FETCH_SIZE / WRITE_SIZE are KB per dispatch (mean); MB = 1e6 bytes",
           "correction": "HBM bytes = 2 x FETCH_SIZE + 1 x WRITE_SIZE (calibration below: a 100.27 MB copy reports FETCH 50.1 MB, WRITE 100.3 MB, for 16 B/lane and for 4 B/lane loads alike)",
           "calibration_on_tools_mem_probe": calib, "frames_per_dispatch": frames, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if "rocclr" in k or "at::" in k or "retile" in k:
            continue
        f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        out["kernels"][k] = {"FETCH_SIZE_KB_per_dispatch": round(fetch.get(k, 0.0), 1), "WRITE_SIZE_KB_per_dispatch": round(write.get(k, 0.0), 1),
                             "hbm_read_MB_per_frame": round(2 * f / frames / 1e6, 2), "hbm_written_MB_per_frame": round(w / frames / 1e6, 2),
                             "hbm_bytes_per_frame": int((2 * f + w) / frames)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
