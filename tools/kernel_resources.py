#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of the kernels, from the remarks the last build left behind (sumcheck_amd/build/*.log; no GPU,
no recompilation).  tools/kernel_resources.py [substring ...] [--exp]   (default: the big-round, tail and finalize kernels)"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exp = "--exp" in sys.argv
pats = [a for a in sys.argv[1:] if not a.startswith("--")] or ["round1_tree_split", "round_tree_split", "tail_rounds", "finalize_mb", "sum_combos", "fix_multi"]
bdir = os.path.join(ROOT, "sumcheck_amd", "build_exp" if exp else "build")
for f in sorted(os.listdir(bdir)):
    if not f.endswith(".log"):
        continue
    name, d = None, {}
    for line in open(os.path.join(bdir, f), errors="replace"):
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            d[name] = {}
        elif name and ":" in t:
            k, v = t.split(":", 1)
            d[name][k.strip()] = v.strip()
    for n, v in d.items():
        if any(p in n for p in pats):
            print(re.sub(r"^_ZN3scd\d+", "", n)[:44].ljust(44), "VGPR", v.get("VGPRs"), "AGPR", v.get("AGPRs"), "scratch", v.get("ScratchSize [bytes/lane]"),
                  "occ", v.get("Occupancy [waves/SIMD]"), "LDS", v.get("LDS Size [bytes/block]"))
