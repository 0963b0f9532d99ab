"""The interactive protocol (IPForMLSumcheck::prove_round called round by round with the verifier's message) on resident tables:
python tools/interactive_time.py [nv ...]  -> us per whole dialogue and per late round, checked against the oracle"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, sumcheck_amd as sc
from oracle import cref
from tests import helpers as H
from sumcheck_amd import _lib
for _a in [a for a in sys.argv if a.startswith("--policy=")]:  # --policy=key=value (A/B runs, e.g. --policy=tail_slices=0)
    _k, _v = _a[len("--policy="):].split("=")
    _lib.set_policy(_k, int(_v))
    sys.argv.remove(_a)
shapes, nt = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
for nv in [int(a) for a in sys.argv[1:]] or [8, 12, 16]:
    tabs = [cref.synth_table(2024, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(2024, 1000, len(shapes))
    chal = cref.synth_table(2024, 2000, nv)
    op = cref.Prover(H.desc_from(nv, shapes, tabs, coefs), threads=4)
    want = [op.prove_round(None if i == 0 else chal[i - 1]) for i in range(nv)]
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    msgs_v = [None] + [sc.VerifierMsg(chal[i]) for i in range(nv - 1)]
    best, late = 1e9, 1e9
    for rep in range(30):
        st.reset()
        t0 = time.perf_counter()
        got = []
        for i in range(nv):
            if i == max(nv - 8, 0):
                t1 = time.perf_counter()
            got.append(sc.IPForMLSumcheck.prove_round(st, msgs_v[i]).evaluations)
        t2 = time.perf_counter()
        best = min(best, t2 - t0)
        late = min(late, (t2 - t1) / min(8, nv))
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    print(f"nv={nv}: interactive dialogue {best*1e6:.1f} us, last rounds {late*1e6:.1f} us each (incl. the Python call)")
