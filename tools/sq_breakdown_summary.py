#!/usr/bin/env python3
"""gpurun_out/sqb_<counter>/ (tools/sq_breakdown.sh) -> profiles/<tag>_sq_breakdown.json: per big round of the last proof, every collected
SQ counter (summed over XCDs) and its share of SQ_WAVE_CYCLES."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2e"
n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 7  # big rounds of one proof (config 3 at nv=24: 7; config 4: nv - 17)
res = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "sqb_*"))):
    if not os.path.isdir(d):
        continue
    c = os.path.basename(d)[4:]
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not cc:
        continue
    agg = {}
    for r in csv.DictReader(open(cc[0])):
        if "k_round" in r["Kernel_Name"] and "tree" in r["Kernel_Name"]:
            agg.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"].split("(")[0], 0.0])[1] += float(r["Counter_Value"])
    ids = sorted(agg)[-n_big:]
    res[c] = [(agg[i][0], agg[i][1]) for i in ids]
out = {"command": "rocprofv3 --pmc <counter> --kernel-trace --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline (one pass per counter)",
       "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts and XCDs; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES", "rounds": []}
n = min(len(v) for v in res.values())
for i in range(n):
    row = {"round": i + 1, "kernel": res["SQ_WAVE_CYCLES"][i][0]}
    wc = res["SQ_WAVE_CYCLES"][i][1]
    for c, v in res.items():
        row[c] = v[i][1]
        if c != "SQ_WAVE_CYCLES" and not c.startswith("SQ_INSTS") and wc:
            row[c + "_share"] = round(v[i][1] / wc, 4)
    out["rounds"].append(row)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_sq_breakdown.json"), "w"), indent=1)
for r in out["rounds"][:n_big]:
    print(r["round"], r["kernel"][-22:], {k[3:-6]: v for k, v in r.items() if k.endswith("_share")}, {k[9:]: round(v / 1e6, 1) for k, v in r.items() if k.startswith("SQ_INSTS")})
