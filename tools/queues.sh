#!/bin/bash
# GPU box: frames in flight vs hardware queues (GPU_MAX_HW_QUEUES), C++ host loop.
P=pixel-art-raytracer_amd/lib/par_pipeline
for q in 4 8 16; do
  for k in 1 2 3 4 5 6 8; do
    echo -n "GPU_MAX_HW_QUEUES=$q inflight=$k  "
    GPU_MAX_HW_QUEUES=$q $P --frames 3000 --inflight $k | cut -c1-200
  done
done
