#!/bin/bash
# SQ_INSTS_VALU / GRBM_GUI_ACTIVE of k_round_tree per round, for two library builds (before / after the shared reductions)
R=$PWD; cd /tmp; export TMPDIR=/tmp
for tag in prev cur; do
  L=$R/sumcheck_amd/libsumcheck_hip.so; [ $tag = prev ] && L=$R/tools/ab/libsumcheck_hip_prev.so
  for C in SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
    SC_LIB_PATH=$L timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq_${tag}_$C -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sq_${tag}_$C.log 2>&1
    find $R/gpurun_out/sq_${tag}_$C -name "*.db" -delete
  done
done
ls $R/gpurun_out/sq_*/
