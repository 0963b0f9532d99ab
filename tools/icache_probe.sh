#!/bin/bash
# Do the big-round kernels (loops of 5-8 thousand unrolled instructions) fit the instruction cache?  rocprofv3 --pmc passes of the SQC instruction
# cache counters over bench.py (2 proofs).  tools/icache_probe.sh ; python tools/sq_breakdown_summary.py <tag> handles the same layout
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -i "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_ICACHE_BUSY[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_BUSY_CY[A-Z_]*\|SQ_ITEMS" | sort -u > $R/gpurun_out/icache_counters.txt
for C in SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/ic_$C -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/ic_$C.log 2>&1
  find $R/gpurun_out/ic_$C -name "*.db" -delete
done
ls -d $R/gpurun_out/ic_* | wc -l
