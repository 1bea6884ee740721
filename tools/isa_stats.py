"""ISA resource summary of the catalog kernels (no GPU needed): compiles every catalog source for gfx950 into a scratch
cache directory and reads each code object's AMDGPU metadata note — VGPRs, SGPRs, spill counts, LDS, scratch — plus the
occupancy (waves per SIMD) those registers allow on gfx950 (512 VGPRs per SIMD lane incl. AGPRs, granule 8).

    python tools/isa_stats.py [--filter SUBSTR] [--csv profiles/rNN_isa_stats.csv]
"""
import argparse
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernels_of(hsaco: str):
    """[(kernel name, {field: value})] from the NT_AMDGPU_METADATA note"""
    txt = subprocess.run([READELF, "--notes", hsaco], capture_output=True, text=True, check=True).stdout
    out, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:                      # first field of a kernel entry
            cur = {"agpr_count": int(m.group(1))}
            out.append(cur)
            continue
        m = re.match(r"\s+\.(\w+):\s+(.+)$", line)
        if m and cur is not None and m.group(1) in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                                     "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size"):
            v = m.group(2).strip()
            cur[m.group(1)] = int(v) if v.isdigit() else v
    return [(k.get("name", "?"), k) for k in out if "vgpr_count" in k]


def waves_per_simd(vgprs: int, agprs: int) -> int:
    total = max(1, (vgprs + agprs + 7) // 8 * 8)       # unified register file, allocation granule 8
    return max(1, min(8, 512 // total))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--csv", default="")
    args = ap.parse_args()
    from qurious_amd import catalog, planning
    rows = []
    for name, src in catalog.catalog_sources():
        if args.filter and args.filter not in name:
            continue
        with tempfile.TemporaryDirectory() as d:
            planning.compile_to_cache(src, d)
            for obj in glob.glob(os.path.join(d, "*.hsaco")):
                for kname, k in kernels_of(obj):
                    rows.append({"catalog entry": name, "kernel": kname, "vgprs": k["vgpr_count"], "agprs": k["agpr_count"],
                                 "sgprs": k.get("sgpr_count", 0), "vgpr_spills": k.get("vgpr_spill_count", 0),
                                 "sgpr_spills": k.get("sgpr_spill_count", 0), "static_lds_bytes": k.get("group_segment_fixed_size", 0),
                                 "scratch_bytes": k.get("private_segment_fixed_size", 0),
                                 "waves_per_simd": waves_per_simd(k["vgpr_count"], k["agpr_count"])})
    cols = ["catalog entry", "kernel", "vgprs", "agprs", "sgprs", "vgpr_spills", "sgpr_spills", "static_lds_bytes", "scratch_bytes", "waves_per_simd"]
    for r in rows:
        print("  ".join(f"{r[c]!s:>{max(len(c), 8)}}" if c not in ("catalog entry", "kernel") else f"{r[c]:<44}"[:44] for c in cols))
    if args.csv:
        with open(args.csv, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=cols)
            w.writeheader()
            w.writerows(rows)


if __name__ == "__main__":
    main()
