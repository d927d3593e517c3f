#!/bin/bash
# GPU box: us per frame of the C++ host loop on graybox, 512^2/64, 1024^2/512, 2048^2/256 and the headline, alone and
# four in flight, for several values of one environment variable. usage: tools/debug/sizes.sh <VAR> "<v1 v2 ...>" [exe]
var=$1; vals=$2; P=${3:-pixel-art-raytracer_amd/lib/par_pipeline}
us() { python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['us_per_frame'])"; }
for v in $vals; do
  for k in 1 4; do
    g=$(env $var=$v $P --scene graybox --frames 4000 --inflight $k --threads $k | head -1 | us)
    s=$(env $var=$v $P --size 512 --prims 64 --frames 4000 --inflight $k --threads $k | head -1 | us)
    m=$(env $var=$v $P --size 1024 --prims 512 --frames 4000 --inflight $k --threads $k | head -1 | us)
    q=$(env $var=$v $P --size 2048 --prims 256 --frames 4000 --inflight $k --threads $k | head -1 | us)
    h=$(env $var=$v $P --size 4096 --prims 1024 --frames 2000 --inflight $k --threads $k | head -1 | us)
    echo "$var=$v inflight $k: graybox $g  512/64 $s  1024/512 $m  2048/256 $q  headline $h"
  done
done
