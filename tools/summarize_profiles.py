#!/usr/bin/env python3
"""Turns what tools/collect_profiles.sh left under gpurun_out/profiles_r01 into the files committed under profiles/:
kernel-trace stats per workload, the raw PMC rows of the headline kernel, and r01_pmc_summary.json (HBM bytes per launch
of qk_filter_agg = FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md §HBM; separate --pmc passes)."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles_r01")
DST = os.path.join(ROOT, "profiles")
ALGO = {"q1_mini": 25, "q1_full": 78}
KERNEL = "qk_filter_agg"


def counter_rows(workload, counter):
    path = os.path.join(SRC, f"pmc_{counter}_{workload}", f"{workload}_counter_collection.csv")
    with open(path) as f:
        return [r for r in csv.DictReader(f) if r["Kernel_Name"] == KERNEL and r["Counter_Name"] == counter]


def main():
    summary = {}
    for wl in ("q1_mini", "q1_full"):
        bench = json.loads(open(os.path.join(SRC, f"bench_{wl}.json")).read().strip().splitlines()[-1])
        rows = bench["config"]["rows_per_gpu"]
        # bytes the kernel reads per row (the SURVEY §8d figure, or less when it skips the offsets of 1-byte Utf8 columns)
        bpr = bench["roofline"].get("kernel_bytes_per_row", ALGO[wl])
        fetch, write = counter_rows(wl, "FETCH_SIZE"), counter_rows(wl, "WRITE_SIZE")
        # keep the launches over the full table (warm-up + timed + verification), drop the small CPU-sample launches
        big = max(int(r["Grid_Size"]) for r in fetch)
        fetch = [r for r in fetch if int(r["Grid_Size"]) == big]
        write = [r for r in write if int(r["Grid_Size"]) == big]
        fkb = sum(float(r["Counter_Value"]) for r in fetch) / len(fetch)
        wkb = sum(float(r["Counter_Value"]) for r in write) / len(write)
        us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fetch) / len(fetch) / 1e3
        hbm = fkb * 1024 * 2 + wkb * 1024
        summary[wl] = {
            "rows": rows, "survey_bytes_per_row": ALGO[wl], "kernel_bytes_per_row": bpr, "algorithmic_bytes": rows * bpr,
            "FETCH_SIZE_KB_raw": fkb, "WRITE_SIZE_KB_raw": wkb,
            "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md §HBM); WRITE_SIZE exact; "
                          "separate --pmc passes",
            "hbm_bytes_per_launch": hbm, "traffic_over_algorithmic": hbm / (rows * bpr), "kernel_us_under_pmc": us,
            "launches": len(fetch),
            "command": "tools/collect_profiles.sh (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py "
                       "--workload ... --steps 3 --warmup 1)"}
        with open(os.path.join(DST, f"r01_{wl}_pmc_counters.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(fetch[0].keys()))
            w.writeheader()
            for r in fetch + write:
                w.writerow(r)
    with open(os.path.join(DST, "r01_pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    for wl, name in (("q1_mini", "q1mini"), ("q1_full", "q1full"), ("q3", "q3")):
        shutil.copy(os.path.join(SRC, f"trace_{wl}", f"{wl}_kernel_stats.csv"), os.path.join(DST, f"r01_{name}_kernel_stats.csv"))
        shutil.copy(os.path.join(SRC, f"bench_{wl}.json"), os.path.join(DST, f"r01_bench_{name}_under_rocprof.json"))
    print(json.dumps({k: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "ratio": v["traffic_over_algorithmic"],
                          "kernel_us": v["kernel_us_under_pmc"]} for k, v in summary.items()}, indent=1))


if __name__ == "__main__":
    sys.exit(main())
