cd $GRAFT_REPO_ROOT
echo "== single process"; python tools/_diag.py 2>&1 | grep -v amdgpu.ids | grep "R0\|RANK" 
for M in hip sleep; do for T in 256 16; do echo "== torchrun 2 ranks mode=$M OMP_NUM_THREADS=$T"; DIAG_MODE=$M OMP_NUM_THREADS=$T python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 tools/_diag.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" ; done; done
