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
import re
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
    cmdline = "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end"
try:
    pmc_cmdline = open(os.path.join(base, "command_pmc.txt")).read().strip()
except OSError:
    pmc_cmdline = cmdline
summary = {
    **bench.source_shas(), "git_head": os.environ.get("GIT_HEAD"),
    "bench_command": pmc_cmdline,
    "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --output-format csv) -- " + pmc_cmdline,
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

# ---- the dominant kernel's duration from `rocprofv3 --kernel-trace --stats` of the same command: what bench.py's roofline.frac is computed from
# (profiles/rocprof_kernel_latest.json, tied to the tree like the traffic file).  `avg_ns` is what the tracked stats CSV yields (every launch
# of the profiled command: sum of TotalDurationNs over the big-round kernels / sum of Calls); `warm` the same from the trace with the
# warm-up proofs of that command dropped; `per_proof_kernel_total_ms` every kernel of a warm proof added up (must fit inside ms_per_step).
stats = list(csv.DictReader(open(os.path.join(base, "stats", "bench_kernel_stats.csv"))))
dom = [r for r in stats if any(x in r["Name"] for x in kern.split("+"))]
calls = sum(int(r["Calls"]) for r in dom)
total_ns = sum(int(r["TotalDurationNs"]) for r in dom)
trace = sorted(csv.DictReader(open(os.path.join(base, "stats", "bench_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
m = re.search(r"--warmup (\d+)", cmdline)
warmup = int(m.group(1)) if m else 0
proofs, cur = [], None  # a proof = everything from one k_round1_tree launch up to the next
for r in trace:
    name = r["Kernel_Name"]
    if "k_round1_tree" in name:
        cur = []
        proofs.append(cur)
    if cur is not None:
        cur.append((name, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
warm = proofs[warmup:]
wd = [d for pr in warm for (n, d) in pr if any(x in n for x in kern.split("+"))]
ksum = {
    **bench.source_shas(), "git_head": os.environ.get("GIT_HEAD"), "bench_command": cmdline,
    "command": "rocprofv3 --kernel-trace --stats --output-format csv -- " + cmdline,
    "stats_csv": f"profiles/{tag}_rocprofv3_kernel_stats.csv", "trace_csv": f"profiles/{tag}_rocprofv3_kernel_trace.csv",
    "kernel": " + ".join(r["Name"].split("(")[0].replace("void ", "") for r in dom),
    "launches": calls, "total_ns": total_ns, "avg_ns": total_ns / max(calls, 1),
    "proofs": len(proofs), "warmup_proofs_dropped": warmup,
    "warm": {"proofs": len(warm), "launches": len(wd), "avg_ns": sum(wd) / max(len(wd), 1)},
    "launches_per_proof": calls / max(len(proofs), 1),
    "per_proof_kernel_total_ms": (sum(d for pr in warm for (_, d) in pr) / max(len(warm), 1)) * 1e-6,
    "per_proof_big_round_kernels_ms": sum(wd) / max(len(warm), 1) * 1e-6,
}
json.dump(ksum, open(os.path.join(out, f"{tag}_rocprof_kernel.json"), "w"), indent=1)
if not no_latest:
    shutil.copy(os.path.join(out, f"{tag}_rocprof_kernel.json"), os.path.join(out, "rocprof_kernel_latest.json"))
print("stats: launches", calls, "avg_ns", ksum["avg_ns"], "warm avg_ns", ksum["warm"]["avg_ns"], "kernel ms per warm proof", ksum["per_proof_kernel_total_ms"])
