"""One-off differential fuzz on a GPU box: sc_poly_evaluate (200 random polynomials) and sc_gkr_prove (120 random instances,
dims 1..15, random non-zero counts, index-ordered or shuffled) against the C oracle.
python tools/fuzz_eval_gkr.py [n_evaluate n_gkr seed]   (progress is printed as it goes: a time limit loses nothing)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sumcheck_amd as sc
from oracle import cref
from tests import helpers as H
n_eval = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_gkr = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 4242)
bad = 0
for c in range(n_eval):  # evaluate
    nv = int(rng.integers(0, 16)); nt = int(rng.integers(1, 40)); K = int(rng.integers(1, 6))
    shapes = [[int(x) for x in rng.integers(0, nt, size=int(rng.integers(1, 6)))] for _ in range(K)]
    tabs = [cref.synth_table(3000 + c, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(3000 + c, 1000, K)
    point = cref.synth_table(3000 + c, 2000, max(nv, 1))[:nv]
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0" if rng.random() < 0.5 else None) if nv > 0 else (None, None)
    if poly is None:
        continue
    got = poly.evaluate(point)
    if not np.array_equal(got, cref.poly_evaluate(H.desc_from(nv, shapes, tabs, coefs), point)):
        bad += 1; print("EVAL MISMATCH", c, nv, nt, shapes)
print("evaluate fuzz mismatches", bad, flush=True)
for c in range(n_gkr):  # GKR
    dim = int(rng.integers(1, 16)); n = 1 << dim
    nnz = int(rng.integers(1, 2 * n + 1))
    space = 1 << (3 * dim)
    idx = np.unique(rng.integers(0, space, size=nnz, dtype=np.uint64))
    if c % 3 == 1:
        idx = idx[rng.permutation(idx.shape[0])]
    vals = cref.synth_table(5000 + c, 1, idx.shape[0]); f2 = cref.synth_table(5000 + c, 2, n); f3 = cref.synth_table(5000 + c, 3, n); g = cref.synth_table(5000 + c, 4, dim)
    f1 = sc.SparseMultilinearExtension(3 * dim, idx, vals)
    pr = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1, sc.DenseMultilinearExtension(dim, f2), sc.DenseMultilinearExtension(dim, f3), g)
    got = np.stack([m.evaluations for m in pr.phase1_sumcheck_msgs + pr.phase2_sumcheck_msgs])
    want = cref.gkr_prove(idx, vals, dim, f2, f3, g, threads=cref.max_threads())
    want = want[0] if isinstance(want, tuple) else want
    if not np.array_equal(got.reshape(-1), np.asarray(want).reshape(-1)[: got.size]):
        bad += 1; print("GKR MISMATCH", c, dim, idx.shape[0])
    if c % 20 == 19:
        print(f"gkr fuzz: {c + 1} cases, mismatches so far {bad}", flush=True)
print("FUZZ2", "OK" if bad == 0 else f"FAILED {bad}")
