#!/bin/bash
# GPU box: SQ counters per kernel (one frame at a time), several --pmc passes. usage: tools/pmc_kernels.sh <tag> [synthetic|floor|graybox] [extra render flags]  (PMC_SETS=valu: the first pass only)
set -e
tag=${1:-x}; what=${2:-synthetic}; xflags=${3:-0}; sets=${PMC_SETS:-all}
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_LDS SQ_WAIT_ANY"; do
  i=$((i+1))
  if [ "$sets" = valu ] && [ $i -gt 1 ]; then break; fi
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o pmc -- python3 tools/frames.py $what 30 $xflags > $out/p$i.log 2> $out/p$i.err || echo "pass $i failed: $set" >> $out/failed.txt
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
json.dump({"source": "rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/frames.py $what 30 (4096x4096, one frame at a time); median per dispatch, summed over the chip", "kernels": res}, open("$out/sq_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
