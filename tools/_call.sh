cd $GRAFT_REPO_ROOT
python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, p.total_memory)"
P=sumcheck_amd/libsumcheck_hip.so
bash tools/ab.sh -r 1 -w rounds -R "25 c4" $P tools/ab/k1_512.so tools/ab/k1_640.so tools/ab/k1_704.so tools/ab/k1_736.so tools/ab/k1_760.so > gpurun_out/r4x_k1_grid_ab2.txt 2>&1
grep -A5 "== rounds" gpurun_out/r4x_k1_grid_ab2.txt
