#!/usr/bin/env python3
"""Wave-instruction counts per kernel launch and workload from rocprofv3 --pmc passes (tools/profile_round.sh):
SQ_WAVES, SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_SMEM, summed over the chip, median over dispatches.
usage: sq_counters.py <prof dir> <commit> > profiles/sq_counters.json   (bench.py prices its issue roofline with it)"""
import collections, csv, glob, json, os, statistics, sys

root, commit = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "unknown commit")
out = {"collected": commit,
       "source": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -- python3 "
                 "tools/frames.py <workload> 30 (one frame at a time); median per dispatch, summed over the chip; MI355X",
       "workloads": {"headline": "4096x4096, 1024 primitives (tools/frames.py synthetic)",
                     "floor": "4096x4096 full floor of 41616 tiles, every pixel covered (tools/frames.py floor)",
                     "graybox": "480x320, the reference's graybox world (tools/frames.py graybox)"}}
for key, what in (("headline", "synthetic"), ("floor", "floor"), ("graybox", "graybox")):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, f"sq_{what}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0]
            if name.startswith("__amd") or name.startswith("at::"):
                continue
            per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[key] = {k: {c: statistics.median(v) for c, v in cs.items()} for k, cs in sorted(per.items())}
print(json.dumps(out, indent=1))
