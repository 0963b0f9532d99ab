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
for _a in [a for a in sys.argv if a.startswith("--policy=")]:  # --policy=key=value: sc_set_policy before anything runs (A/B runs, e.g. --policy=wide_tree=0)
    _k, _v = _a[len("--policy="):].split("=")
    _lib.set_policy(_k, int(_v))
    sys.argv.remove(_a)
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


def gkr(dim, reps=7):
    """BASELINE config 5: GKRRoundSumcheck::prove, inputs HBM-resident (the library reads them in place) and, for comparison, from
    host memory (H2D inside the call).  roofline: HBM; algorithmic bytes of the whole call (SURVEY 8d style, 32-byte elements):
    two sumcheck phases over two dense tables each, 32 * 2 * (4 * 2^dim - 6) per phase; initialisation: f1 read (40 B per non-zero:
    8 B index + 32 B value) once per sparse fold plus one result write (2 folds), the two eq tables written once (2^dim each), f3 read
    for the scatter terms (nnz' gathers), h_g / f1(g,u,.) written once each, f2 read once (evaluate) and f3 read + written once (scale).
    sc_gkr_prove groups the terms by target cell and adds them in LDS (no sorts); stage times come from SC_GKR_TRACE=1."""
    rng = np.random.default_rng(SEED)
    n = 1 << dim
    idx = np.unique(rng.integers(0, 1 << (3 * dim), size=2 * n, dtype=np.uint64))[:n]
    vals, f2, f3, g = cref.synth_table(SEED, 1, idx.shape[0]), cref.synth_table(SEED, 2, n), cref.synth_table(SEED, 3, n), cref.synth_table(SEED, 4, dim)
    nnz = int(idx.shape[0])

    def run(f1, m2, m3):
        ts = []
        for i in range(reps + 3):
            t0 = time.perf_counter(); pr = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1, m2, m3, g); dt = time.perf_counter() - t0
            if i >= 3:
                ts.append(dt)
        return pr, ts

    _, th = run(sc.SparseMultilinearExtension(3 * dim, idx, vals), sc.DenseMultilinearExtension(dim, f2), sc.DenseMultilinearExtension(dim, f3))
    td = lambda a: torch.from_numpy(a.view(np.int64)).to(dev)
    prd, tdv = run(sc.SparseMultilinearExtension(3 * dim, td(idx), td(vals)), sc.DenseMultilinearExtension(dim, td(f2)), sc.DenseMultilinearExtension(dim, td(f3)))
    t0 = time.perf_counter(); want, _ = cref.gkr_prove(idx, vals, dim, f2, f3, g, threads=cref.max_threads()); tc = time.perf_counter() - t0
    assert np.array_equal(np.stack([m.evaluations for m in prd.phase1_sumcheck_msgs]), want[0])
    assert np.array_equal(np.stack([m.evaluations for m in prd.phase2_sumcheck_msgs]), want[1])
    alg = 2 * 32 * 2 * (4 * n - 6) + 2 * (40 * nnz + 40 * nnz) + 2 * 32 * n + 32 * nnz + 2 * 32 * n + 32 * n + 2 * 32 * n
    med = float(np.median(tdv))
    return {"dim": dim, "nnz": nnz, "gpu_ms_median_device_resident": 1e3 * med, "gpu_ms_min_device_resident": 1e3 * min(tdv),
            "gpu_ms_median_host_inputs_incl_h2d": 1e3 * float(np.median(th)), "cpu_port_s": tc, "cpu_threads": cref.max_threads(),
            "speedup_device_resident": tc / med,
            "roofline": {"bound": "hbm", "algorithmic_bytes": alg, "achieved": alg / med / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / med / 1e9 / 8000.0,
                         "note": "latency-bound: 40 sumcheck rounds over 2^20-entry tables (about 25 us each) and a dozen short initialisation kernels"}}


def streamed(nv, shapes, nt, reps=3):
    """out-of-core mode: tables in PINNED host memory, streamed through HBM in rounds 1 and 2 (sc_prover_init_streamed)"""
    host = []
    for u in range(nt):
        t = torch.empty((1 << nv, 4), dtype=torch.int64, device=dev)
        _lib.check(sc.lib().sc_synth_table_device(SEED, u, 0, 1 << nv, C.c_void_p(t.data_ptr())))
        h = torch.empty((1 << nv, 4), dtype=torch.int64, pin_memory=True)
        h.copy_(t)
        host.append(h)
        del t
    torch.cuda.empty_cache()
    coefs = cref.synth_table(SEED, 1000, len(shapes))
    mles = [sc.DenseMultilinearExtension(nv, h) for h in host]
    poly = sc.ListOfProductsOfPolynomials(nv)
    for k, sh in enumerate(shapes):
        poly.add_product([mles[i] for i in sh], coefs[k])
    st = sc.IPForMLSumcheck.prover_init(poly, streamed_chunk_log2=0)
    ts = []
    for i in range(reps + 1):
        st.reset()
        t0 = time.perf_counter(); st.prove(); dt = time.perf_counter() - t0
        if i:
            ts.append(dt)
    U, D = nt, max(len(s) for s in shapes) + 1
    ops = ((1 << nv) - 1) * sum(2 * len(s) * D + len(s) + D for s in shapes) + 3 * U * ((1 << nv) - 2)
    med = float(np.median(ts))
    return {"nv": nv, "shapes": shapes, "tables_GiB": U * (1 << nv) * 32 / 2**30, "gpu_ms_median": 1e3 * med, "field_ops_per_s": ops / med,
            "pcie_GBps": 2 * U * (1 << nv) * 32 / med / 1e9, "note": "the input crosses PCIe twice (rounds 1 and 2); PCIe-bound"}


def wide(nv, ms, reps=7):
    """the reference's own test shapes (ml_sumcheck/test.rs:122-167: products of 4..12 multiplicands over fresh tables), at a size where
    they matter: every product over its own tables.  Next to the time: algorithmic GB/s (32 U (4 2^nv - 6)), the reference
    algorithm's multiplications per second and the products the kernels execute per second (a product of M multiplicands at M + 1 nodes:
    (M - 1)(M + 1) products per pair + M binds per pair of the next round) against the 160 G/s carry-free ceiling."""
    shapes, nt = [], 0
    for m in ms:
        shapes.append(list(range(nt, nt + m)))
        nt += m
    r = ml(nv, shapes, nt, reps=reps, cpu=False)
    med = r["gpu_ms_median"] * 1e-3
    D = max(ms) + 1
    r["reference_muls_per_s"] = ((1 << nv) - 1) * sum(m * D for m in ms) / med + nt * ((1 << nv) - 2) / med
    tree = {1: 0, 2: 3, 3: 7, 4: 11}

    def products(m):  # Montgomery-product equivalents per pair: the product tree (<= 4), the halves' trees + one product per node + the
        if m <= 4:    # extensions at half a product each (5..8, kernels_wide.hip), node by node beyond that
            return tree[m]
        if m <= 8:
            mb = m - 4
            ext = (m + 1 - 5) + (m + 1 - (mb + 1 if mb >= 2 else 3))
            return 11 + tree[mb] + (m + 1) + 0.5 * ext
        if m <= 12:   # 9..12 (kernels_wide16.hip): the first eight as above at nine nodes, the rest's tree, one product per node, extensions
            mb = m - 8
            ext = (m + 1 - 9) + (m + 1 - (mb + 1 if mb >= 2 else 3))
            return (11 + 11 + 9 + 0.5 * 8) + tree[mb] + (m + 1) + 0.5 * ext
        return (m - 1) * (m + 1)
    exe = sum((products(m) + m * 97 / 153) for m in ms) * ((1 << nv) - 1)
    r["executed_products_per_s"] = exe / med
    r["frac_of_fe_mul_ceiling"] = exe / med / 160e9
    r["frac_of_hbm_peak"] = r["algorithmic_GBps"] / 8000.0
    r["multiplicands"] = ms
    r["kernels"] = ("k_round1_tree_split / k_round_tree_split (every product <= 4)" if max(ms) <= 4 else
                    "k_prod_tree (<= 4) + k_prod_tree_wide<M> (5..8: halves' trees, node extension), one launch per product" if max(ms) <= 8 else
                    "k_fix_multi (bind pass) + k_prod_tree_wide16<M> (9..12: a tree of the trees)" if max(ms) <= 12 and _lib.get_policy("wide_tree") != 0 else "k_fix per table + k_sum_generic per product")
    return r


if "--only-wide12" in sys.argv:  # for rocprofv3 runs of one product of twelve alone
    print(json.dumps({"one_product_of_12": wide(20, [12])}, indent=1))
    sys.exit(0)
if "--wide" in sys.argv:  # the M >= 5 paths, timed
    nvw = 20
    print(json.dumps({"test_normal_shape_nv20_5_products_of_4_to_8": wide(nvw, [4, 5, 6, 7, 8]),
                      "five_products_of_5": wide(nvw, [5, 5, 5, 5, 5]),
                      "five_products_of_8": wide(nvw, [8, 8, 8, 8, 8]),
                      "one_product_of_12": wide(nvw, [12]),
                      "one_product_of_9": wide(nvw, [9]),
                      "test_normal_shape_nv20_5_products_of_4_to_12": wide(nvw, [4, 6, 8, 10, 12]),
                      "one_product_of_6": wide(nvw, [6]),
                      "for_scale_five_products_of_4": wide(nvw, [4, 4, 4, 4, 4])}, indent=1))
    sys.exit(0)
if "--only-streamed" in sys.argv:
    print(json.dumps({"streamed_nv26_3tables": streamed(26, [[0, 1, 2]], 3)}, indent=1))
    sys.exit(0)
if "--only-gkr" in sys.argv:  # for rocprofv3 runs of config 5 alone
    print(json.dumps({"config5_gkr": gkr(20)}, indent=1))
    sys.exit(0)
out = {"config2": ml(20, [[0, 1, 2]], 3), "readme_bench_shape": ml(20, [[0, 1, 2], [3, 4, 5]], 6), "config5_gkr": gkr(20)}
if "--config4" in sys.argv:  # the whole nv=28 job of config 4 on ONE GPU (24 GiB of tables + 13.5 GiB of bound-table buffers in HBM)
    out["config4_nv28_one_gpu"] = ml(28, [[0, 1, 2]], 3, reps=3, cpu=False)
    out["config4_shard_nv25_one_gpu"] = ml(25, [[0, 1, 2]], 3, reps=5, cpu=False)
print(json.dumps(out, indent=1))
