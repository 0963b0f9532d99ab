#!/usr/bin/env python3
"""gpurun_out/sq_{prev,cur}_{SQ_INSTS_VALU,GRBM_GUI_ACTIVE}/ (tools/sq_counters.sh) -> profiles/<tag>_sq_counters.json:
per big round of the last proof, wave-level VALU instructions, duration and shader clock for both settings."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2e"
def rounds(setting, counter):
    d = os.path.join(ROOT, "gpurun_out", f"sq_{setting}_{counter}")
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt))}
    agg = {}
    for r in csv.DictReader(open(cc)):
        if "k_round" not in r["Kernel_Name"] or "tree" not in r["Kernel_Name"]:
            continue
        agg.setdefault(r["Dispatch_Id"], [r["Kernel_Name"].split("(")[0], 0.0])[1] += float(r["Counter_Value"])
    ids = sorted(agg, key=int)[-7:]  # the last proof's seven big rounds
    return [(agg[i][0], agg[i][1], dur.get(i)) for i in ids]
out = {"command": "rocprofv3 --pmc <counter> --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline (experiments build; one pass per counter and setting)",
       "settings": {"prev": "SC_SPLIT=0: every block walks all products (k_round1_tree, k_round_tree)", "cur": "one product per block row (k_round1_tree_split, k_round_tree_split)"},
       "note": "counters summed over the 8 XCDs; GRBM_GUI_ACTIVE / 8 / duration = shader clock; device-side waits are off under counter collection", "rounds": []}
v = {s: rounds(s, "SQ_INSTS_VALU") for s in ("prev", "cur")}
g = {s: rounds(s, "GRBM_GUI_ACTIVE") for s in ("prev", "cur")}
for i in range(7):
    row = {"round": i + 1}
    for s in ("prev", "cur"):
        row[s] = {"kernel": v[s][i][0], "SQ_INSTS_VALU": v[s][i][1], "dur_us": v[s][i][2], "GRBM_GUI_ACTIVE": g[s][i][1],
                  "shader_clock_GHz": g[s][i][1] / 8 / (g[s][i][2] * 1e3) if g[s][i][2] else None}
    row["valu_ratio_cur_over_prev"] = row["cur"]["SQ_INSTS_VALU"] / row["prev"]["SQ_INSTS_VALU"]
    out["rounds"].append(row)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters_split.json"), "w"), indent=1)
for r in out["rounds"]:
    print(r["round"], {s: (round(r[s]["SQ_INSTS_VALU"] / 1e6, 1), round(r[s]["dur_us"], 1), round(r[s]["shader_clock_GHz"] or 0, 2)) for s in ("prev", "cur")})
