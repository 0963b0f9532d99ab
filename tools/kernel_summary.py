#!/usr/bin/env python3
"""rocprofv3 kernel_stats.csv -> the short per-kernel table kept under profiles/ (python tools/kernel_summary.py <stats.csv> "<header line>")"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
print(f"{'kernel':48s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'%':>7s}")
for r in rows:
    n = r["Name"]
    if "rocprim" in n:
        short = "rocprim::" + (n.split("wrapped_")[1].split("<")[0] if "wrapped_" in n else n.split("rocprim::")[-1][:30])
    else:
        m = re.search(r"(scd::|gkr::)?k_\w+(<[^>(]*>)?", n)
        short = m.group(0) if m else n.split("(")[0]
    print(f"{short[:48]:48s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e3:11.1f} {float(r['AverageNs'])/1e3:10.2f} {100*float(r['TotalDurationNs'])/tot:7.2f}")
