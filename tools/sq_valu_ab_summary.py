#!/usr/bin/env python3
"""gpurun_out/sqv_<tag>_<name>/ (tools/sq_valu_ab.sh) -> profiles/<tag>_sq_insts_valu.json: wave-level VALU instructions and duration of
the last proof's seven big rounds for each library build (counter summed over the 8 XCDs; device-side waits are off under counter collection)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, names = sys.argv[1], sys.argv[2:]
def rounds(name):
    d = os.path.join(ROOT, "gpurun_out", f"sqv_{tag}_{name}")
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt))}
    agg = {}
    for r in csv.DictReader(open(cc)):
        if "k_round" not in r["Kernel_Name"] or "tree" not in r["Kernel_Name"]:
            continue
        agg.setdefault(r["Dispatch_Id"], [r["Kernel_Name"].split("(")[0].replace("void ", ""), 0.0])[1] += float(r["Counter_Value"])
    ids = sorted(agg, key=int)[-7:]
    return [{"kernel": agg[i][0], "SQ_INSTS_VALU": agg[i][1], "dur_us": dur.get(i)} for i in ids]
out = {"command": "rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (SC_LIB_PATH = each build)",
       "builds": {n: rounds(n) for n in names}}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_sq_insts_valu.json"), "w"), indent=1)
for i in range(7):
    print(i + 1, {n: (round(out["builds"][n][i]["SQ_INSTS_VALU"] / 1e6, 1), round(out["builds"][n][i]["dur_us"], 1)) for n in names})
