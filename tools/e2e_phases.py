#!/usr/bin/env python3
"""tools/e2e_phases.py [nv]: where the host-tables-in, proof-out path of config 3 spends its time (bench.py's config.end_to_end in pieces):
the bare copy, the copy in the staged initialisation's chunk pattern, IPForMLSumcheck::prover_init over host tables (staged / plain) and
the proof that follows it, the one-shot MLSumcheck::prove."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sumcheck_amd as sc
from sumcheck_amd import _lib
import ctypes as C

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 24
shapes, U = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
dev = torch.device("cuda", 0)
n = 1 << nv
tables = []
for u in range(U):
    t = torch.empty((n, 4), dtype=torch.int64, device=dev)
    _lib.check(sc.lib().sc_synth_table_device(0x5C20241008, u, 0, n, C.c_void_p(t.data_ptr())))
    tables.append(t)
ct = torch.empty((len(shapes), 4), dtype=torch.int64, device=dev)
_lib.check(sc.lib().sc_synth_table_device(0x5C20241008, 1000, 0, len(shapes), C.c_void_p(ct.data_ptr())))
coefs = ct.cpu().numpy().view(np.uint64)
host = [t.cpu().pin_memory() for t in tables]
scratch = [torch.empty_like(t) for t in tables]


def wall(f, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize(dev)
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def bare():
    for h, d in zip(host, scratch):
        d.copy_(h, non_blocking=True)


def chunked(levels=4):
    for c in range(levels + 1):
        sh = c + 1 if c < levels else levels
        first, cnt = n - (n >> c), n >> sh
        for h, d in zip(host, scratch):
            d[first:first + cnt].copy_(h[first:first + cnt], non_blocking=True)


print("bare H2D, 10 copies            %.3f ms" % wall(bare))
print("H2D in 5 x 10 geometric chunks %.3f ms" % wall(chunked))
print("H2D in 8 x 10 equal chunks     %.3f ms" % wall(lambda: [scratch[u][c * (n // 8):(c + 1) * (n // 8)].copy_(host[u][c * (n // 8):(c + 1) * (n // 8)], non_blocking=True) for c in range(8) for u in range(U)]))
del scratch
mles = [sc.DenseMultilinearExtension(nv, h) for h in host]
poly = sc.ListOfProductsOfPolynomials(nv)
for k, sh in enumerate(shapes):
    poly.add_product([mles[i] for i in sh], coefs[k])
for staged in (1, 0, 1, 0):
    with _lib.policy(staged_init=staged):
        ti, tp, tt = [], [], []
        for _ in range(4):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            st = sc.IPForMLSumcheck.prover_init(poly)
            t1 = time.perf_counter()
            st.prove()
            t2 = time.perf_counter()
            st.close()
            ti.append((t1 - t0) * 1e3); tp.append((t2 - t1) * 1e3); tt.append((t2 - t0) * 1e3)
        one = wall(lambda: sc.MLSumcheck.prove(poly), reps=4)
        print("staged_init=%d: prover_init %.3f ms, prove after it %.3f ms, together %.3f ms (best of 4; a fresh handle each time); one-shot MLSumcheck.prove %.3f ms"
              % (staged, min(ti), min(tp), min(tt), one))
