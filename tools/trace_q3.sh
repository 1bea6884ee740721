#!/bin/bash
# rocprofv3 kernel trace of the Q3 bench (run on the GPU box through gpurun, from the repo root):
#   tools/trace_q3.sh <out_dir> [bench args...]     prints the per-launch durations of the last query's kernels
set -o pipefail
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o q3 -- python3 bench.py --workload q3 --sf 10 --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - $OUT <<'PY'
import csv, sys, glob, json
out = sys.argv[1]
print(open(out + "/bench.json").read()[:400])
rows = list(csv.DictReader(open(glob.glob(out + "/**/q3_kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last query = everything after the third-last qk_filter_agg launch's end
agg = [i for i, r in enumerate(rows) if r["Kernel_Name"] == "qk_filter_agg"]
first = agg[-2] + 1 if len(agg) >= 2 else 0
tot = 0.0
t_begin = int(rows[first]["Start_Timestamp"])
for r in rows[first:agg[-1] + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{(int(r['Start_Timestamp']) - t_begin) / 1e3:9.1f} us  {d:8.1f} us  {r['Kernel_Name'][:70]}")
print(f"kernel time of one query: {tot:.1f} us; span {(int(rows[agg[-1]]['End_Timestamp']) - t_begin) / 1e3:.1f} us")
PY
