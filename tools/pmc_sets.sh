#!/bin/bash
# GPU box: arbitrary --pmc counter sets per kernel, one frame at a time (tools/frames.py).
# usage: tools/pmc_sets.sh <tag> <workload> <set1> [<set2> ...]   each set: counters separated by spaces, quoted
tag=$1; what=$2; shift 2
out=gpurun_out/pmcs_$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python3 tools/frames.py $what 20 0 > $out/p$i.log 2> $out/p$i.err || echo "pass $i failed: $set" >> $out/failed.txt
done
python3 - <<PY
import csv, glob, json, statistics, collections
res = collections.defaultdict(dict)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0]
        if name.startswith("__amd") or name.startswith("at::"):
            continue
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in per.items():
        for c, v in cs.items():
            res[k][c] = statistics.median(v)
json.dump({"source": "rocprofv3 --kernel-trace --pmc <set> -- python3 tools/frames.py $what 20; median per dispatch, summed over the chip", "kernels": res}, open("$out/counters.json", "w"), indent=1)
for k, v in res.items():
    print(k, json.dumps(v))
PY
