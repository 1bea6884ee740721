#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun, from the repo root):
#   kernel-trace stats of the three benchmark workloads, and FETCH_SIZE / WRITE_SIZE of the headline kernel in
#   separate --pmc passes (MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 128-B requests at 64 B on gfx950 -> x2).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r01
mkdir -p $OUT
for wl in "q1_mini" "q1_full --rows 59986052" "q3 --sf 10"; do
  name=$(echo $wl | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$name -o $name -- python3 bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "trace $name exit $?"
done
for c in FETCH_SIZE WRITE_SIZE; do
  for wl in "q1_mini" "q1_full --rows 59986052"; do
    name=$(echo $wl | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_$name -o $name -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_${c}_$name.err
    echo "pmc $c $name exit $?"
  done
done
