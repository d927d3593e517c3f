#!/bin/bash
# GPU box: the dense floor scene with parts of the frame switched off (flag bits 24-28; the output is then wrong by design)
P=pixel-art-raytracer_amd/lib/par_pipeline
run() { echo -n "$1: "; $P --scene floor --frames 600 --inflight 4 --threads 4 --flags $2 | grep -o '"us_per_frame": [0-9.]*'; }
run "all                        " 0
run "no stores (25)             " $((1<<25))
run "no shading/shadow (26)     " $((1<<26))
run "no shading, no stores      " $(((1<<25)|(1<<26)))
run "no primary (24) (=> no hit)" $((1<<24))
run "no walks (27)              " $((1<<27))
run "no walks, no pixel work    " $(((1<<27)|(1<<24)|(1<<25)|(1<<26)))
run "no fill (28)               " $((1<<28))
