#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/profile_round.sh).
usage: hbm_traffic.py <prof dir> > profiles/hbm_traffic.json
Counter values are KiB; gfx950 FETCH_SIZE counts half of wide (16 B per lane) reads, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section). Median over dispatches."""
import collections, csv, glob, json, os, statistics, sys

root = sys.argv[1]
collected = sys.argv[2] if len(sys.argv) > 2 else "unknown commit"
raw = collections.defaultdict(dict)
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv for {counter} under {root}")
    per = collections.defaultdict(list)
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            name = name.split("<")[0]
            if name.startswith("__amd_rocclr") or name.startswith("at::"):
                continue
            per[name].append(float(r["Counter_Value"]))
    for name, v in per.items():
        raw[name][counter] = statistics.median(v)
out = {"collected": collected,
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 "
                 "tools/frames.py synthetic 40 (4096x4096, 1024 primitives); median over dispatches; MI355X",
       "units": "counter values are KiB; hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE counts half "
                "of wide 16-B/lane reads, MI355X_MICROARCH.md HBM section)",
       "raw_kib": {k: raw[k] for k in sorted(raw)}}
for k in sorted(raw):
    out[k + "_hbm_bytes_per_launch"] = int((2 * raw[k].get("FETCH_SIZE", 0.0) + raw[k].get("WRITE_SIZE", 0.0)) * 1024)
print(json.dumps(out, indent=1))
