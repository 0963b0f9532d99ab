timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -x -q -m gpu -k "golden or random_shapes or gkr or give_up" 2>&1 | tail -3
SC_HOST_TRACE=1 timeout 120 python - <<'PY' 2>&1 | tail -22
import os, ctypes as C, numpy as np, torch, time
import sumcheck_amd as sc
from sumcheck_amd import _lib
nv=24; shapes=[[0,1,2,3],[4,5,6],[7,8],[9]]
dev=torch.device("cuda:0"); tabs=[]
for u in range(10):
    t=torch.empty((1<<nv,4),dtype=torch.int64,device=dev); _lib.check(sc.lib().sc_synth_table_device(1,u,0,1<<nv,C.c_void_p(t.data_ptr()))); tabs.append(t)
from oracle import cref
coefs=cref.synth_table(1,1000,4)
mles=[sc.DenseMultilinearExtension(nv,t) for t in tabs]
poly=sc.ListOfProductsOfPolynomials(nv)
for k,sh in enumerate(shapes): poly.add_product([mles[i] for i in sh],coefs[k])
st=sc.IPForMLSumcheck.prover_init(poly,borrow=True)
for _ in range(3): st.reset(); st.prove()
import sys; sys.stderr.flush(); print("=====", flush=True)
st.reset(); t0=time.perf_counter(); st.prove(); print("proof ms", (time.perf_counter()-t0)*1e3)
PY
timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
timeout 200 python tools/bench_configs.py 2>/dev/null | grep -E "gpu_ms"
