#!/bin/bash
# GPU box: par_pipeline of several builds (build/<name>/) x several values of one environment variable, alternating.
# usage: tools/debug/abenv.sh <reps> <VAR> "<v1 v2 ...>" <name1> [name2 ...] [-- par_pipeline args]
reps=$1; var=$2; vals=$3; shift 3
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" = "--" ] && shift
args=("$@")
[ ${#args[@]} -eq 0 ] && args=(--size 4096 --prims 1024 --frames 4000 --inflight 4 --threads 4)
for i in $(seq $reps); do
  for v in $vals; do
    line="$var=$v:"
    for n in "${names[@]}"; do
      exe=build/$n/par_pipeline; [ "$n" = tree ] && exe=pixel-art-raytracer_amd/lib/par_pipeline
      x=$(env $var=$v $exe "${args[@]}" | grep us_per_frame | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['us_per_frame'])")
      line="$line $n $x"
    done
    echo "$line"
  done
done
