"""Repeat whole proofs on resident tables and compare every one with the oracle's (pipelined late rounds: thousands of
mailbox hand-overs).  python tools/soak.py  -> "SOAK OK"."""
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
print("SOAK", "OK" if bad == 0 else f"FAILED {bad}")
