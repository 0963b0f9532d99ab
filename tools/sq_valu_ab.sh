#!/bin/bash
# SQ_INSTS_VALU (wave-level VALU instructions) and duration of the big-round kernels per round for several library builds (same box):
# tools/sq_valu_ab.sh TAG name1=path1.so name2=path2.so ...  -> gpurun_out/sqv_TAG_<name>/ ; python tools/sq_valu_ab_summary.py TAG name1 name2 ...
R=$PWD; TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
for kv in "$@"; do
  name=${kv%%=*}; lib=${kv#*=}
  SC_LIB_PATH=$R/$lib timeout 240 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/sqv_${TAG}_$name -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sqv_${TAG}_$name.log 2>&1
  find $R/gpurun_out/sqv_${TAG}_$name -name "*.db" -delete
done
