#!/usr/bin/env python3
"""Timings of the other BASELINE configs on one GPU (reported in DESIGN.md; bench.py stays on the headline config):
config 2 (nv=20, 1 product of 3), the README-bench shape (nv=20, 2 x 3), config 5 (GKR dim=20), each next to the CPU port."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sumcheck_amd as sc
from oracle import cref
from sumcheck_amd import _lib

SEED = 0x5C20241008
dev = torch.device("cuda:0")


def ml(nv, shapes, nt, reps=10, cpu=True):
    tabs = []
    for u in range(nt):
        t = torch.empty((1 << nv, 4), dtype=torch.int64, device=dev)
        _lib.check(sc.lib().sc_synth_table_device(SEED, u, 0, 1 << nv, C.c_void_p(t.data_ptr())))
        tabs.append(t)
    coefs = cref.synth_table(SEED, 1000, len(shapes))
    mles = [sc.DenseMultilinearExtension(nv, t) for t in tabs]
    poly = sc.ListOfProductsOfPolynomials(nv)
    for k, sh in enumerate(shapes):
        poly.add_product([mles[i] for i in sh], coefs[k])
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    for _ in range(3):
        st.reset(); st.prove()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        st.reset()
        t0 = time.perf_counter(); st.prove(); ts.append(time.perf_counter() - t0)
    if not cpu:  # too large for a host copy: GPU time only, field-ops from SURVEY 8d's formula
        U, D = nt, max(len(s) for s in shapes) + 1
        ops = ((1 << nv) - 1) * sum(2 * len(s) * D + len(s) + D for s in shapes) + 3 * U * ((1 << nv) - 2)
        return {"nv": nv, "shapes": shapes, "gpu_ms_median": 1e3 * float(np.median(ts)), "gpu_ms_min": 1e3 * min(ts), "field_ops": ops,
                "gpu_field_ops_per_s": ops / float(np.median(ts)), "algorithmic_GBps": 32 * U * (4 * (1 << nv) - 6) / float(np.median(ts)) / 1e9}
    host = [t.cpu().numpy().view(np.uint64) for t in tabs]
    d = cref.PolyDesc(nv, [(coefs[k], sh) for k, sh in enumerate(shapes)], host)
    t0 = time.perf_counter(); cref.ml_prove(d, threads=cref.max_threads()); tc = time.perf_counter() - t0
    ops = d.field_ops()
    return {"nv": nv, "shapes": shapes, "gpu_ms_median": 1e3 * float(np.median(ts)), "gpu_ms_min": 1e3 * min(ts), "field_ops": ops,
            "gpu_field_ops_per_s": ops / float(np.median(ts)), "cpu_port_s": tc, "cpu_threads": cref.max_threads(), "speedup": tc / float(np.median(ts))}


def gkr(dim, reps=5):
    rng = np.random.default_rng(SEED)
    n = 1 << dim
    idx = np.unique(rng.integers(0, 1 << (3 * dim), size=2 * n, dtype=np.uint64))[:n]
    vals, f2, f3, g = cref.synth_table(SEED, 1, idx.shape[0]), cref.synth_table(SEED, 2, n), cref.synth_table(SEED, 3, n), cref.synth_table(SEED, 4, dim)
    f1 = sc.SparseMultilinearExtension(3 * dim, idx, vals)
    m2, m3 = sc.DenseMultilinearExtension(dim, f2), sc.DenseMultilinearExtension(dim, f3)
    ts = []
    for i in range(reps + 2):
        t0 = time.perf_counter(); sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1, m2, m3, g); dt = time.perf_counter() - t0
        if i >= 2:
            ts.append(dt)
    t0 = time.perf_counter(); cref.gkr_prove(idx, vals, dim, f2, f3, g, threads=cref.max_threads()); tc = time.perf_counter() - t0
    return {"dim": dim, "nnz": int(idx.shape[0]), "gpu_ms_median_incl_h2d": 1e3 * float(np.median(ts)), "gpu_ms_min": 1e3 * min(ts), "cpu_port_s": tc,
            "cpu_threads": cref.max_threads(), "speedup": tc / float(np.median(ts))}


out = {"config2": ml(20, [[0, 1, 2]], 3), "readme_bench_shape": ml(20, [[0, 1, 2], [3, 4, 5]], 6), "config5_gkr": gkr(20)}
if "--config4" in sys.argv:  # the whole nv=28 job of config 4 on ONE GPU (24 GiB of tables + 13.5 GiB of bound-table buffers in HBM)
    out["config4_nv28_one_gpu"] = ml(28, [[0, 1, 2]], 3, reps=3, cpu=False)
    out["config4_shard_nv25_one_gpu"] = ml(25, [[0, 1, 2]], 3, reps=5, cpu=False)
print(json.dumps(out, indent=1))
