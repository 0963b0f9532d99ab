#!/bin/bash
# Shader clock per big-round kernel: GRBM_GUI_ACTIVE (and SQ_BUSY_CYCLES, SQ_WAVE_CYCLES) per dispatch divided by the dispatch's duration.
# tools/kernel_clock.sh TAG [bench.py arguments, e.g. --config 4 --nv 25]   -> gpurun_out/TAG_kernel_clock.txt
TAG=$1; shift
R=$PWD; cd /tmp; export TMPDIR=/tmp
for C in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES; do
  rm -rf $R/gpurun_out/kc_$C
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/kc_$C -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --min-gpu-seconds 0 $* > $R/gpurun_out/kc_$C.log 2>&1
  find $R/gpurun_out/kc_$C -name "*.db" -delete
done
cd $R; python - "$TAG" <<'PY' > gpurun_out/${TAG}_kernel_clock.txt
import csv, glob, sys, collections
out = collections.OrderedDict()
for C in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES"):
    fs = glob.glob(f"gpurun_out/kc_{C}/**/*counter_collection.csv", recursive=True)
    if not fs: print("no csv for", C); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == C and ("k_round" in r["Kernel_Name"])]
    for i, r in enumerate(rows):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else 0
        out.setdefault(i, {"kernel": r["Kernel_Name"][:40], "grid": r.get("Grid_Size")})[C] = (float(r["Counter_Value"]), dur)
print("# per k_round* dispatch (both proofs of the run): counter value, duration in us, value / duration (per ns)")
for i, d in out.items():
    s = f"{i:3d} {d['kernel']:<42} grid {d['grid']:>8}"
    for C, vd in d.items():
        if C in ("kernel", "grid"): continue
        v, dur = vd
        s += f" | {C} {v:.4g} {dur/1e3:.1f}us {v/max(dur,1):.3f}"
    print(s)
PY
cat gpurun_out/${TAG}_kernel_clock.txt | head -50
