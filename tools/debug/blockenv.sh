#!/bin/bash
# GPU box: steady rate and the time of a block of 20 frames (started on an idle device) for several values of one
# environment variable. usage: tools/debug/blockenv.sh <reps> <VAR> "<v1 v2 ...>" [exe]
reps=$1; var=$2; vals=$3; exe=${4:-pixel-art-raytracer_amd/lib/par_pipeline}
for i in $(seq $reps); do
  for v in $vals; do
    s=$(env $var=$v $exe --size 4096 --prims 1024 --frames 4000 --inflight 4 --threads 4 | grep -o '"us_per_frame": [0-9.]*' | grep -o '[0-9.]*$')
    b=$(env $var=$v $exe --size 4096 --prims 1024 --frames 400 --inflight 4 --block 20 | head -1 | grep -o '= [0-9.]* us per frame')
    a=$(env $var=$v $exe --size 4096 --prims 1024 --frames 2000 --inflight 1 | grep -o '"us_per_frame": [0-9.]*' | grep -o '[0-9.]*$')
    echo "$var=$v: steady $s  block of 20 $b  alone $a"
  done
done
