"""Per-launch durations of the LAST `n` kernel launches of a rocprofv3 kernel trace CSV (gaps between launches included).
    python tools/trace_last.py <dir> [n]
"""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"gap {((s - prev) / 1e3 if prev else 0):8.1f} us  dur {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:80]}")
    prev = e
