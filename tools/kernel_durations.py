#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel and launch geometry, dispatch count and median / mean /
min / max duration in microseconds. usage: kernel_durations.py <dir or *_kernel_trace.csv> [title]"""
import csv, glob, os, statistics, sys

path = sys.argv[1]
if os.path.isdir(path):
    found = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    if not found:
        sys.exit("no *kernel_trace.csv under " + path)
    path = found[-1]
if len(sys.argv) > 2:
    print(sys.argv[2])
groups = {}
with open(path) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if name.startswith("void "):
            name = name[5:]
        grid = int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1) or 1) * int(row.get("Grid_Size_Z", 1) or 1)
        key = f"{name[:40]}|grid={grid}|wg={row['Workgroup_Size_X']}"
        groups.setdefault(key, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print("per kernel and launch geometry: dispatches, median / mean / min / max duration in microseconds")
for key, d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    print(f"{key:70s} n={len(d):5d} median {statistics.median(d):9.2f} mean {statistics.fmean(d):9.2f} "
          f"min {min(d):9.2f} max {max(d):9.2f}")

# the launch chain: what precedes each kernel on the device (gap = its start - the previous dispatch's end, any queue)
rows = []
with open(path) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), name[:40]))
rows.sort()
gaps = {}
for prev, cur in zip(rows, rows[1:]):
    gaps.setdefault((prev[2], cur[2]), []).append((cur[0] - prev[1]) / 1e3)
print("gaps between consecutive dispatches (previous end -> next start), pairs seen at least 20 times: median us")
for (a, b), g in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(g) >= 20:
        print(f"{a:40s} -> {b:40s} n={len(g):5d} median gap {statistics.median(g):7.2f}")
