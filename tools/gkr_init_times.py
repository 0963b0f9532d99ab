"""initialize_phase_one / initialize_phase_two as stand-alone calls (sc_gkr_phase_one / _two) on device-resident inputs:
python tools/gkr_init_times.py [dim]   -> median ms of 10 calls each, for an index-ordered and for a shuffled list, checked against the oracle"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sumcheck_amd as sc
from oracle import cref
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << dim
rng = np.random.default_rng(5)
idx = np.unique(rng.integers(0, 1 << (3 * dim), size=2 * n, dtype=np.uint64))[:n]
vals, f3, g, u = cref.synth_table(5, 1, idx.shape[0]), cref.synth_table(5, 3, n), cref.synth_table(5, 4, dim), cref.synth_table(5, 6, dim)
wh, wi, wv = cref.gkr_phase_one(idx, vals, dim, f3, g)
wgu = cref.gkr_phase_two(wi, wv, dim, u)
td = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
m3 = sc.DenseMultilinearExtension(dim, td(f3))
for name, perm in (("index-ordered", None), ("shuffled", rng.permutation(idx.shape[0]))):
    i2, v2 = (idx, vals) if perm is None else (idx[perm], vals[perm])
    f1 = sc.SparseMultilinearExtension(3 * dim, td(i2), td(v2))
    t1, t2 = [], []
    for rep in range(13):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h_g, f1_g = sc.initialize_phase_one(f1, m3, g)
        t1.append(time.perf_counter() - t0)
        f1g_in = f1_g if perm is None else sc.SparseMultilinearExtension(2 * dim, f1_g.indices.flip(0).contiguous(), f1_g.values.flip(0).contiguous())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f1_gu = sc.initialize_phase_two(f1g_in, u)
        t2.append(time.perf_counter() - t0)
    ok = (np.array_equal(h_g.evaluations.cpu().numpy().view(np.uint64), wh) and np.array_equal(f1_g.indices.cpu().numpy().view(np.uint64), wi)
          and np.array_equal(f1_g.values.cpu().numpy().view(np.uint64), wv) and np.array_equal(f1_gu.evaluations.cpu().numpy().view(np.uint64), wgu))
    print(f"dim {dim}, {idx.shape[0]} non-zeros, {name} list: initialize_phase_one {1e3*np.median(t1[3:]):.3f} ms, initialize_phase_two {1e3*np.median(t2[3:]):.3f} ms, {'bit-exact' if ok else 'MISMATCH'}")
