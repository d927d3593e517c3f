  a=$($A --size 4096 --prims 1024 --inflight 4 --threads 4 --block 20 | grep -o "= [0-9.]* us per frame")
  b=$($B --size 4096 --prims 1024 --inflight 4 --threads 4 --block 20 | grep -o "= [0-9.]* us per frame")
  echo "headline block of 20: base $a  tree $b"
#!/bin/bash
# GPU box: the same par_pipeline run with two builds of the library, alternating (A = build/base, the round's start;
# B = the tree's), so that box-to-box differences cancel. usage: tools/debug/ab.sh [reps] [-- extra par_pipeline args]
reps=${1:-3}
A=build/base/par_pipeline
B=pixel-art-raytracer_amd/lib/par_pipeline
us() { "$@" | head -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.readline())['us_per_frame'])"; }
for i in $(seq $reps); do
  for k in 4 1; do
    a=$(us $A --size 4096 --prims 1024 --frames 4000 --inflight $k --threads $k)
    b=$(us $B --size 4096 --prims 1024 --frames 4000 --inflight $k --threads $k)
    echo "headline inflight $k: base $a  tree $b"
  done
  a=$($A --size 4096 --prims 1024 --inflight 4 --threads 4 --block 20 | grep -o "= [0-9.]* us per frame")
  b=$($B --size 4096 --prims 1024 --inflight 4 --threads 4 --block 20 | grep -o "= [0-9.]* us per frame")
  echo "headline block of 20: base $a  tree $b"
done
