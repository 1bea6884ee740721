#!/bin/bash
# Q3 SF10 under rocprofv3 for several settings of the join layout switches (run on the GPU box through gpurun):
#   tools/q3_variants.sh <out_dir> "NAME=ENV1=v,ENV2=v" ...      per variant: the last query's per-kernel durations
set -o pipefail
OUT=$1; shift
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  ( IFS=,; for kv in $envs; do [ "$kv" != "-" ] && export "$kv"; done
    timeout -k 10 240 bash tools/trace_q3.sh $OUT/$name > $OUT/$name.txt 2>&1 )
  echo "== $name ($envs): $(grep -h 'kernel time of one query' $OUT/$name.txt)"
  grep -h "qk_join\|k_gather_fixed\|fillBuffer" $OUT/$name.txt | awk '{printf "   %8s us  %s\n", $3, $5}' | head -12
done
