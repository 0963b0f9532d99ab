#!/bin/bash
# SQ_INSTS_VALU / GRBM_GUI_ACTIVE of the big-round kernels per round, for two settings of the experiments build (default: the previous
# kernels, SC_SPLIT=0, against one product per block row).  tools/sq_counters.sh ; then python tools/sq_counters_summary.py
R=$PWD; cd /tmp; export TMPDIR=/tmp
export SC_LIB_VARIANT=exp
for tag in prev cur; do
  S=1; [ $tag = prev ] && S=0
  for C in SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
    SC_SPLIT=$S timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq_${tag}_$C -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sq_${tag}_$C.log 2>&1
    find $R/gpurun_out/sq_${tag}_$C -name "*.db" -delete
  done
done
ls $R/gpurun_out/sq_*/
