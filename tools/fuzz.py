"""One-off differential fuzz on a GPU box: random product lists (shared tables, repeated factors, 1..12 multiplicands, up to 14
products) at sizes that run big rounds (merged and per-product launches), whole proofs against the C oracle; a third of the cases as
the interactive dialogue (prove_round per round, the oracle transcript's challenges, random pauses beyond the resident kernel's patience).
python tools/fuzz.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sumcheck_amd as sc
from oracle import cref
from tests import helpers as H

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
bad = 0
t0 = time.time()
for c in range(cases):
    nv = int(rng.choice([1, 2, 5, 9, 13, 16, 17, 18, 19], p=[.04, .04, .05, .07, .1, .1, .2, .25, .15]))
    nt = int(rng.integers(33, 49)) if rng.random() < 0.1 else int(rng.integers(1, 9))  # (a tenth of the lists over more tables than a launch's arguments hold)
    K = int(rng.integers(1, 15)) if rng.random() < 0.15 else int(rng.integers(1, 6))
    um = rng.random()
    maxm = 12 if um < 0.2 else 8 if um < 0.4 else 4
    shapes = [[int(x) for x in rng.integers(0, nt, size=int(rng.integers(1, maxm + 1)))] for _ in range(K)]
    tabs = [cref.synth_table(1000 + c, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(1000 + c, 1000, K)
    want, wrand = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=cref.max_threads())
    dev = "cuda:0" if rng.random() < 0.5 else None
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device=dev)
    if rng.random() < 0.33:
        st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
        got = []
        for i in range(nv):
            if rng.random() < 0.1:
                time.sleep(0.003)
            got.append(sc.IPForMLSumcheck.prove_round(st, None if i == 0 else sc.VerifierMsg(wrand[i - 1])).evaluations)
        got = np.stack(got)
        ok = np.array_equal(got, want) and np.array_equal(st.randomness, wrand[: nv - 1])
        st.close()
    else:
        proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly)
        got = np.stack([m.evaluations for m in proof])
        ok = np.array_equal(got, want) and np.array_equal(state.randomness, wrand)
    if not ok:
        bad += 1
        print("MISMATCH", c, nv, nt, shapes, dev)
    if c % 20 == 19:
        print(f"fuzz: {c + 1} cases, mismatches so far {bad}, {time.time() - t0:.0f} s", flush=True)
print(f"FUZZ {'OK' if bad == 0 else 'FAILED'}: {cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
