"""Latency of small whole proofs on resident tables (the persistent tail kernel / pipelined rounds): python tools/small_proofs.py [nv ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sumcheck_amd as sc
from oracle import cref
from tests import helpers as H
SHAPES = {"c3": ([[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10), "c2": ([[0, 1, 2]], 3), "gkr": ([[0, 1]], 2), "shared": ([[0, 1, 2], [1, 3]], 4)}  # SC_SHAPE selects (default c3)
shapes, nt = SHAPES[os.environ.get("SC_SHAPE", "c3")]
out = []
for nv in [int(a) for a in sys.argv[1:]] or [6, 10, 12, 14, 16]:
    tabs = [cref.synth_table(2024, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(2024, 1000, len(shapes))
    want, _ = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=cref.max_threads())
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(400):
            st.reset()
            proof = st.prove()
        best = min(best, (time.perf_counter() - t0) / 400)
    assert np.array_equal(np.asarray(proof).reshape(want.shape), want)
    out.append(f"nv={nv}: {best*1e6:.1f} us")
print("  ".join(out))
