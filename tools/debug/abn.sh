#!/bin/bash
# GPU box: par_pipeline of several builds (build/<name>/par_pipeline; "tree" = the tree's own), alternating, so that
# box-to-box differences cancel. usage: tools/debug/abn.sh <reps> <name1> [name2 ...] [-- par_pipeline args]
reps=$1; shift
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" = "--" ] && shift
args=("$@")
[ ${#args[@]} -eq 0 ] && args=(--size 4096 --prims 1024 --frames 4000 --inflight 4 --threads 4)
for i in $(seq $reps); do
  line=""
  for n in "${names[@]}"; do
    exe=build/$n/par_pipeline; [ "$n" = tree ] && exe=pixel-art-raytracer_amd/lib/par_pipeline
    v=$($exe "${args[@]}" | head -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['us_per_frame'])")
    line="$line $n $v"
  done
  echo "$line"
done
