#!/usr/bin/env python3
"""Summarises the CSVs that tools/pmc_passes.sh left under <dir> for <tag>:

    python tools/summarize_pmc.py <dir> <tag> [<out.json>]

Per (kernel, grid size): number of launches, mean duration (us) and the mean of every collected counter per launch.
HBM bytes per launch = FETCH_SIZE (KB) x 1024 x 2 + WRITE_SIZE (KB) x 1024: on gfx950 FETCH_SIZE tallies the 128-byte
requests of a wide streaming read at 64 bytes (MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact; the two come from
separate --pmc passes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("qhip::", "")
    return name.split("(")[0]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    acc = defaultdict(lambda: defaultdict(list))   # (kernel, grid) -> counter -> values
    dur = defaultdict(list)
    for d in sorted(glob.glob(os.path.join(src, f"pmc_{tag}_*"))):
        if not os.path.isdir(d):
            continue
        for path in glob.glob(os.path.join(d, "**", f"{tag}_counter_collection.csv"), recursive=True):
            seen = set()
            with open(path) as f:
                for r in csv.DictReader(f):
                    key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r["Dispatch_Id"] not in seen:
                        seen.add(r["Dispatch_Id"])
                        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    kernels = {}
    for key, counters in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]])):
        k = {"kernel": key[0], "grid_size": key[1], "launches_seen": len(dur[key]), "mean_us_under_pmc": sum(dur[key]) / max(1, len(dur[key]))}
        for name, vals in sorted(counters.items()):
            k[name] = sum(vals) / len(vals)
        if "FETCH_SIZE" in k or "WRITE_SIZE" in k:
            k["hbm_bytes_per_launch"] = k.get("FETCH_SIZE", 0.0) * 1024 * 2 + k.get("WRITE_SIZE", 0.0) * 1024
        if "TCC_HIT_sum" in k and "TCC_MISS_sum" in k and k["TCC_HIT_sum"] + k["TCC_MISS_sum"] > 0:
            k["l2_hit_rate"] = k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
        kernels[f"{key[0]}@{key[1]}"] = k
    out = {"tag": tag, "correction": "gfx950: FETCH_SIZE x2 (128-B requests tallied at 64 B), WRITE_SIZE exact; separate --pmc passes "
                                      "(MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots)", "kernels": kernels}
    text = json.dumps(out, indent=1)
    if out_path:
        with open(out_path, "w") as f:
            f.write(text + "\n")
    # a compact view on stdout: the ten longest kernels
    for name, k in list(kernels.items())[:12]:
        keys = [c for c in k if c not in ("kernel", "grid_size")]
        print(name, {c: (round(k[c], 3) if isinstance(k[c], float) else k[c]) for c in keys})


if __name__ == "__main__":
    main()
