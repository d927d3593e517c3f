#!/bin/bash
# GPU box: kernel durations, one frame at a time and pipelined (C++ host), quickly. usage: tools/prof_quick.sh <tag> [workload]
tag=${1:-q}; what=${2:-synthetic}
out=gpurun_out/pq_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/one -o one -- python3 tools/frames.py $what 300 > $out/one.log 2>&1
python3 tools/kernel_durations.py $out/one "one frame at a time ($what)" > $out/one_kernel_durations.txt
rocprofv3 --kernel-trace --output-format csv -d $out/pipe -o pipe -- pixel-art-raytracer_amd/lib/par_pipeline --frames 1500 --inflight 4 > $out/pipe.log 2>&1
python3 tools/kernel_durations.py $out/pipe "4 frames in flight (C++ host)" > $out/pipe_kernel_durations.txt
python3 tools/timeline.py $(find $out/pipe -name "*kernel_trace.csv") 0.4 0.7 > $out/pipe_timeline.txt
head -12 $out/one_kernel_durations.txt; head -12 $out/pipe_kernel_durations.txt; cat $out/pipe_timeline.txt; cat $out/pipe.log | tail -2
