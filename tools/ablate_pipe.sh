#!/bin/bash
# GPU box: frames in flight with parts of the frame switched off (flag bits 24-28; the output is then wrong by design)
P=pixel-art-raytracer_amd/lib/par_pipeline
run() { echo -n "$1: "; $P --frames 3000 --inflight ${3:-4} --threads ${3:-4} --flags $2 | grep -o '"us_per_frame": [0-9.]*'; }
for k in 4; do
echo "--- inflight $k"
run "all                      " 0 $k
run "no fill (28)             " $((1<<28)) $k
run "no walks (27)            " $((1<<27)) $k
run "no primary/shading/stores" $(((1<<24)|(1<<25)|(1<<26))) $k
run "no stores (25)           " $((1<<25)) $k
run "no fill, no walks        " $(((1<<28)|(1<<27))) $k
run "no fill, no pixel work   " $(((1<<28)|(1<<24)|(1<<25)|(1<<26))) $k
run "no fill/walks/pixel work " $(((1<<28)|(1<<27)|(1<<24)|(1<<25)|(1<<26))) $k
run "fill only (no walks/pixel)" $(((1<<27)|(1<<24)|(1<<25)|(1<<26))) $k
done
