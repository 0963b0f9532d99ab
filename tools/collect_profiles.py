#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of gpurun_out/prof_<tag>/ into profiles/ and derive the HBM traffic per launch of the
dominant kernel.  Usage: python tools/collect_profiles.py <tag> [kernel substring, default k_round_tree] [--no-latest]
(--no-latest: a profile of another workload -- config 4 -- does not become profiles/hbm_traffic_latest.json, which bench.py's default
run reads).  The summary records the tree it was taken on: content hashes of the kernel sources and of bench.py (what bench.py
compares before it reports `traffic`) and GIT_HEAD from the environment when the caller passed it (a GPU box has no .git)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = [a for a in sys.argv[1:] if a != "--no-latest"]
no_latest = "--no-latest" in sys.argv
tag = argv[0]
kern = argv[1] if len(argv) > 1 else "k_round_tree+k_round1_tree"
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_shas)
base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
out = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(base, "stats", "bench_kernel_stats.csv"), os.path.join(out, f"{tag}_rocprofv3_kernel_stats.csv"))
shutil.copy(os.path.join(base, "stats", "bench_kernel_trace.csv"), os.path.join(out, f"{tag}_rocprofv3_kernel_trace.csv"))
counters = {}
for name, f in (("FETCH_SIZE", "pmc_fetch/bench_counter_collection.csv"), ("WRITE_SIZE", "pmc_write/bench_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(os.path.join(base, f))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    counters[name] = {k: {"calls": n, "sum_KB": v, "per_call_KB": v / n} for k, (n, v) in agg.items()}
# the merged big-round launch is two kernels: k_round1_tree (round 1) and k_round_tree (rounds 2..7): both count
keys = [k for k in counters["FETCH_SIZE"] if any(x in k for x in kern.split("+"))]
key = " + ".join(keys)
f = {"calls": sum(counters["FETCH_SIZE"][k]["calls"] for k in keys), "sum_KB": sum(counters["FETCH_SIZE"][k]["sum_KB"] for k in keys)}
w = {"calls": sum(counters["WRITE_SIZE"][k]["calls"] for k in keys), "sum_KB": sum(counters["WRITE_SIZE"][k]["sum_KB"] for k in keys)}
try:
    cmdline = open(os.path.join(base, "command.txt")).read().strip()
except OSError:
    cmdline = "python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
summary = {
    **bench.source_shas(), "git_head": os.environ.get("GIT_HEAD"),
    "bench_command": cmdline,
    "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline",
    "corrections": "FETCH_SIZE x2 (MI355X_MICROARCH.md: gfx950 counts 64 B per 128-B request for 16 B/lane coalesced reads; checked on the "
                   "round-1 launch of k_prod_tree<4>: 4 x 2^24 x 32 B = 2097152 KB read, counter 1048848 KB); WRITE_SIZE x1 (calibrated on k_synth: 10 x 2^24 x 32 B "
                   "written, counter 5242880 KB)",
    "kernel": key,
    "launches": f["calls"],
    "traffic_bytes_per_launch": (2 * f["sum_KB"] + w["sum_KB"]) * 1024 / f["calls"],
    "algorithmic_bytes_per_launch_big_rounds_only": None,
    "counters": counters,
}
json.dump(summary, open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w"), indent=1)
if not no_latest:
    shutil.copy(os.path.join(out, f"{tag}_hbm_traffic.json"), os.path.join(out, "hbm_traffic_latest.json"))
print(key, "launches", f["calls"], "traffic/launch", summary["traffic_bytes_per_launch"])
