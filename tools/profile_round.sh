#!/bin/bash
# GPU box: the round's rocprofv3 evidence. usage: tools/profile_round.sh <tag>   (writes gpurun_out/prof_<tag>/)
set -e
tag=${1:-x}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 1000 --warmup 100 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $out/bench_profiled.json 2> $out/rocprof_bench.err
python3 tools/kernel_durations.py $out/bench "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline" > $out/bench_kernel_durations.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/one -o one -- python3 tools/frames.py synthetic 300 > $out/one.log 2> $out/rocprof_one.err
python3 tools/kernel_durations.py $out/one "rocprofv3 --kernel-trace --stats -- python3 tools/frames.py synthetic 300 (one frame at a time)" > $out/one_kernel_durations.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o pmc -- python3 tools/frames.py synthetic 40 > $out/pmc_$c.log 2> $out/rocprof_pmc_$c.err
done
find $out -name "*stats*.csv" | head; find $out -name "*counter*" | head
python3 tools/hbm_traffic.py $out > $out/hbm_traffic.json
