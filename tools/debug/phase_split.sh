#!/bin/bash
# GPU box: instructions per wavefront of the render kernel with phases of the frame switched off (instrumented
# instantiation; flag bits 24 no primary pass, 26 no shading / shadow test, 27 no shadow walks = empty walk lists).
# usage: tools/debug/phase_split.sh <tag> [workload] [kernel name part]   (default: floor, render_tiles)
tag=${1:-x}; what=${2:-floor}; kern=${3:-render_tiles}
export TMPDIR=/tmp
for f in 0 $((1<<27)) $((1<<26)) $((1<<24)) $(((1<<24)|(1<<26))); do
  out=gpurun_out/phase_${tag}_$f
  mkdir -p $out
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $out -o pmc -- python3 tools/frames.py $what 20 $((f|(1<<29))) > $out/log 2> $out/err
  python3 - $out $f $kern <<'PY'
import csv, glob, sys, statistics, collections
out, f, kern = sys.argv[1], int(sys.argv[2]), sys.argv[3]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if kern in r["Kernel_Name"]:
            per[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
dur = []
for fn in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if kern in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
m = {k: statistics.median(v["v"]) for k, v in per.items()}
w = m.get("SQ_WAVES", 1)
print(f"flags {f:#x}: waves {w:.0f} " + " ".join(f"{k[9:]}/wave {m[k]/w:.1f}" for k in sorted(m) if k != "SQ_WAVES") + f" kernel_us {statistics.median(dur):.1f}")
PY
done
