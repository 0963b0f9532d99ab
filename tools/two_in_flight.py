#!/usr/bin/env python3
"""Throughput with SEVERAL proofs in flight on one GPU: T prover handles on T host threads over the same borrowed config-3 tables (nv=24),
each proving back to back.  A single proof leaves the GPU nearly idle for its 17 latency-bound rounds (~10 % of its time); another
proof's big rounds can run there.  python tools/two_in_flight.py [nv] [proofs per thread] [threads ...]   (default 24 60 1 2 3)"""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, sumcheck_amd as sc
from sumcheck_amd import _lib
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
counts = [int(a) for a in sys.argv[3:]] or [1, 2, 3]
shapes = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]
dev = torch.device("cuda:0")
tabs = []
for u in range(10):
    t = torch.empty((1 << nv, 4), dtype=torch.int64, device=dev)
    _lib.check(sc.lib().sc_synth_table_device(0x5C20241008, u, 0, 1 << nv, C.c_void_p(t.data_ptr())))
    tabs.append(t)
ct = torch.empty((4, 4), dtype=torch.int64, device=dev)
_lib.check(sc.lib().sc_synth_table_device(0x5C20241008, 1000, 0, 4, C.c_void_p(ct.data_ptr())))
coefs = ct.cpu().numpy().view(np.uint64)
torch.cuda.synchronize()


def make():
    mles = [sc.DenseMultilinearExtension(nv, t) for t in tabs]
    poly = sc.ListOfProductsOfPolynomials(nv)
    for k, sh in enumerate(shapes):
        poly.add_product([mles[i] for i in sh], coefs[k])
    return sc.IPForMLSumcheck.prover_init(poly, borrow=True)


ref = None
for T in counts:
    states = [make() for _ in range(T)]
    for st in states:  # warm up, and every handle's proof is the same proof
        st.reset()
        p = np.asarray(st.prove())
        if ref is None:
            ref = p
        assert np.array_equal(p, ref)
    bar = threading.Barrier(T + 1)
    bad = [0] * T

    def work(i):
        st = states[i]
        _lib.check(sc.lib().sc_set_device(0))
        bar.wait()
        for _ in range(reps):
            st.reset()
            if not np.array_equal(np.asarray(st.prove()), ref):
                bad[i] += 1
        bar.wait()

    ths = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for t in ths:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in ths:
        t.join()
    print(f"nv={nv} threads={T}: {T * reps} proofs in {dt * 1e3:.1f} ms -> {dt / (T * reps) * 1e3:.3f} ms per proof (aggregate), mismatches {sum(bad)}")
    del states
