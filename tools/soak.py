"""Repeat whole proofs on resident tables and compare every one with the oracle's (persistent tail kernel and pipelined late rounds:
tens of thousands of mailbox hand-overs, grid barriers and block retirements).  Also two provers proving concurrently from two
threads on the same GPU.  python tools/soak.py  -> "SOAK OK"."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sumcheck_amd as sc
from oracle import cref
from tests import helpers as H
bad = 0
for nv, shapes, nt, reps in ((14, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10, 3000), (18, [[0, 1, 2], [1, 3]], 4, 600), (6, [[0, 1]], 2, 3000)):
    tabs = [cref.synth_table(2024, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(2024, 1000, len(shapes))
    want, wrand = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=cref.max_threads())
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    t0 = time.perf_counter()
    for i in range(reps):
        st.reset()
        proof = st.prove()
        got = np.stack([m.evaluations for m in proof]) if isinstance(proof, list) else np.asarray(proof)
        if not np.array_equal(got.reshape(want.shape), want):
            bad += 1
    print(f"nv={nv}: {reps} proofs, {1e3*(time.perf_counter()-t0)/reps:.3f} ms each, mismatches so far {bad}")
# two handles, two threads, one GPU: their persistent kernels and wait kernels share the device
import threading
def worker(nv, shapes, nt, reps, out, k):
    tabs = [cref.synth_table(3000 + k, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(3000 + k, 1000, len(shapes))
    want, _ = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=4)
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    b = 0
    for i in range(reps):
        st.reset()
        if not np.array_equal(np.asarray(st.prove()).reshape(want.shape), want):
            b += 1
    out[k] = b
out = [None, None]
ts = [threading.Thread(target=worker, args=(15, [[0, 1, 2], [3]], 4, 1500, out, 0)), threading.Thread(target=worker, args=(19, [[0, 1, 2, 3], [1, 2]], 4, 300, out, 1))]
t0 = time.perf_counter()
for t in ts: t.start()
for t in ts: t.join()
print(f"two concurrent provers: mismatches {out}, {time.perf_counter()-t0:.1f} s")
bad += sum(x or 0 for x in out) + sum(1 for x in out if x is None)
print("SOAK", "OK" if bad == 0 else f"FAILED {bad}")
