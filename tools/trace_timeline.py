"""Timeline of one steady-state query from a rocprofv3 --kernel-trace --memory-copy-trace run: every kernel / copy in order,
with its duration and the idle gap in front of it (the median-length occurrence between two launches of the first kernel).
usage: python tools/trace_timeline.py <dir> <first-kernel-substr> [launches of that kernel per query]"""
import csv
import glob
import sys

d, first = sys.argv[1], sys.argv[2]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
starts = [i for i, e in enumerate(ev) if first in e[2]]
spans = sorted((ev[starts[k + per]][0] - ev[starts[k]][0], k) for k in range(0, len(starts) - per, per))
if not spans:
    sys.exit("not enough occurrences of the first kernel")
k = spans[len(spans) // 3][1]
a, b = starts[k], starts[k + per]
prev_end = ev[a - 1][1] if a else ev[a][0]
busy = 0
for s, e, n in ev[a:b]:
    print(f"{(s - ev[a][0]) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {n}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"query: {(ev[b][0] - ev[a][0]) / 1e3:.1f} us wall, {busy / 1e3:.1f} us busy, {b - a} launches")
