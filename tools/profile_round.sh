#!/bin/bash
# GPU box: the round's rocprofv3 evidence. usage: tools/profile_round.sh <tag> [commit]   (writes gpurun_out/prof_<tag>/)
set -e
tag=${1:-x}
commit=${2:-unknown}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 200 --warmup 50 > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -o bench -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras > $out/bench_profiled.json 2> $out/rocprof_bench.err
python3 tools/kernel_durations.py $out/bench "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras" > $out/bench_kernel_durations.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/one -o one -- python3 tools/frames.py synthetic 300 > $out/one.log 2> $out/rocprof_one.err
python3 tools/kernel_durations.py $out/one "rocprofv3 --kernel-trace --stats -- python3 tools/frames.py synthetic 300 (one frame at a time)" > $out/one_kernel_durations.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/floor -o floor -- python3 tools/frames.py floor 60 > $out/floor.log 2> $out/rocprof_floor.err
python3 tools/kernel_durations.py $out/floor "rocprofv3 --kernel-trace --stats -- python3 tools/frames.py floor 60 (one frame at a time, full floor)" > $out/floor_kernel_durations.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o pmc -- python3 tools/frames.py synthetic 40 > $out/pmc_$c.log 2> $out/rocprof_pmc_$c.err
done
python3 tools/hbm_traffic.py $out "$commit" > $out/hbm_traffic.json
for w in synthetic floor graybox; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $out/sq_$w -o pmc -- python3 tools/frames.py $w 30 > $out/sq_$w.log 2> $out/rocprof_sq_$w.err
done
python3 tools/sq_counters.py $out "$commit" > $out/sq_counters.json
find $out -name "*stats*.csv" | head
