"""MLSumcheck::prove as a one-shot call (sc_ml_prove: build the prover, prove, free it) against the same proof on a kept handle:
python tools/oneshot_time.py [nv ...]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sumcheck_amd as sc
from sumcheck_amd import _lib
from oracle import cref
shapes, nt = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
for nv in [int(a) for a in sys.argv[1:]] or [16, 20, 24]:
    mles = []
    for s in range(nt):
        t = torch.empty((1 << nv, 4), dtype=torch.int64, device="cuda:0")
        _lib.check(sc.lib().sc_synth_table_device(77, s, 0, 1 << nv, C.c_void_p(t.data_ptr())))
        mles.append(sc.DenseMultilinearExtension(nv, t))
    coefs = cref.synth_table(77, 1000, len(shapes))
    poly = sc.ListOfProductsOfPolynomials(nv)
    for k, sh in enumerate(shapes):
        poly.add_product([mles[i] for i in sh], coefs[k])
    one, kept = [], []
    for rep in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        proof = sc.MLSumcheck.prove(poly)
        one.append(time.perf_counter() - t0)
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    for rep in range(7):
        st.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p2 = st.prove()
        kept.append(time.perf_counter() - t0)
    same = np.array_equal(np.stack([m.evaluations for m in proof]).reshape(-1), np.asarray(p2).reshape(-1))
    print(f"nv={nv}: one-shot MLSumcheck.prove {1e3*np.median(one[2:]):.3f} ms, kept handle {1e3*np.median(kept[2:]):.3f} ms, same proof: {same}")
