"""sc_poly_evaluate (ListOfProductsOfPolynomials::evaluate) at config 3's size on resident tables: python tools/evaluate_time.py [nv]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sumcheck_amd as sc
from sumcheck_amd import _lib
from oracle import cref
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 24
shapes, nt = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
dev = torch.device("cuda:0")
mles = []
for s in range(nt):
    t = torch.empty((1 << nv, 4), dtype=torch.int64, device=dev)
    _lib.check(sc.lib().sc_synth_table_device(77, s, 0, 1 << nv, C.c_void_p(t.data_ptr())))
    mles.append(sc.DenseMultilinearExtension(nv, t))
coefs = cref.synth_table(77, 1000, len(shapes))
poly = sc.ListOfProductsOfPolynomials(nv)
for k, sh in enumerate(shapes):
    poly.add_product([mles[i] for i in sh], coefs[k])
point = cref.synth_table(77, 2000, nv)
ts = []
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    val = poly.evaluate(point)
    ts.append(time.perf_counter() - t0)
print(f"evaluate nv={nv}, {nt} tables: {1e3*np.median(ts[2:]):.3f} ms (min {1e3*min(ts[2:]):.3f}); value limb0 {int(np.asarray(val).reshape(-1)[0]):#x}")
