#!/bin/bash
# Where the big-round kernels' wave cycles go: one rocprofv3 --pmc pass per SQ counter over bench.py (2 proofs), production library.
# tools/sq_breakdown.sh [extra bench.py arguments, e.g. --config 4 --nv 25] ; python tools/sq_breakdown_summary.py <tag>
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/sqb_*
for C in SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM_WR SQ_ACTIVE_INST_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sqb_$C -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --min-gpu-seconds 0 $* > $R/gpurun_out/sqb_$C.log 2>&1
  find $R/gpurun_out/sqb_$C -name "*.db" -delete
done
ls -d $R/gpurun_out/sqb_* | wc -l
