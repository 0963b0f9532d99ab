"""Two provers proving concurrently from two threads on one GPU (python tools/concurrent_provers.py [nv_a nv_b reps])."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sumcheck_amd as sc
from oracle import cref
from tests import helpers as H
nv_a = int(sys.argv[1]) if len(sys.argv) > 1 else 15
nv_b = int(sys.argv[2]) if len(sys.argv) > 2 else 19
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
def worker(nv, shapes, nt, reps, out, k):
    try:
        tabs = [cref.synth_table(3000 + k, s, 1 << nv) for s in range(nt)]
        coefs = cref.synth_table(3000 + k, 1000, len(shapes))
        want, _ = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=4)
        poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
        st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
        b, worst = 0, 0.0
        for i in range(reps):
            st.reset()
            t0 = time.perf_counter()
            got = np.asarray(st.prove())
            worst = max(worst, time.perf_counter() - t0)
            if not np.array_equal(got.reshape(want.shape), want):
                b += 1
        out[k] = (b, round(worst * 1e3, 2))
    except Exception as e:
        out[k] = repr(e)[:200]
out = [None, None]
ts = [threading.Thread(target=worker, args=(nv_a, [[0, 1, 2], [3]], 4, reps, out, 0)), threading.Thread(target=worker, args=(nv_b, [[0, 1, 2, 3], [1, 2]], 4, max(reps // 4, 1), out, 1))]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
print(f"concurrent provers nv={nv_a},{nv_b}: (mismatches, worst proof ms) = {out}, {time.perf_counter()-t0:.1f} s")
