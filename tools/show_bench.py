import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"],4))
rec=d.get("records",{})
for name,r in rec.items():
    if "kernels" in r:
        for k in r["kernels"]:
            print("   ", k["operator"], k["kernel"], round(k["kernel_ms"]*1e3,1), "us", "frac", round(k["frac"],3), k.get("groups",""))
    if "cases" in r:
        for c,v in r["cases"].items(): print("   ", c, round(v["ms_per_call"],4), "ms", "device", round(v["device_ms"],4), "frac", round(v["roofline"]["frac"],3))
if "aggregate_table" in d: print("   ", d["aggregate_table"])
