./tools/instr_bench.bin 2>&1 | tail -12 > gpurun_out/r3f_instr_latency.txt; cat gpurun_out/r3f_instr_latency.txt
for sh in c3 gkr; do SC_SHAPE=$sh SC_LIB_PATH=$PWD/tools/ab/tail_clocks.so timeout 200 python tools/tail_clocks.py 12 2>&1 | grep -v amdgpu; done | tee gpurun_out/r3f_tail_clocks.txt
