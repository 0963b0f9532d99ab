"""Shared test helpers: golden fixture loading and building identical inputs for the oracle and the HIP path."""
import glob
import json
import os

import numpy as np

from oracle import cref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hx(s: str) -> int:
    return int(s, 16)


def load(name: str) -> dict:
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def ml_cases():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "ml_*.json")))


def gkr_cases():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "gkr_*.json")))


def mont(vals) -> np.ndarray:
    return cref.ints_to_mont([hx(v) if isinstance(v, str) else v for v in vals])


def golden_tables(case: dict):
    """all tables of a golden ML case as Montgomery limb arrays (by original table id)"""
    return [mont(t) for t in case["tables"]]


def oracle_desc(case: dict) -> cref.PolyDesc:
    tabs = golden_tables(case)
    flat = [tabs[i] for i in case["flattened_table_ids"]]
    prods = [(mont([c])[0], ix) for c, ix in case["products"]]
    return cref.PolyDesc(case["nv"], prods, flat)


def hip_poly(case: dict, device=None):
    """build a sumcheck_amd.ListOfProductsOfPolynomials from a golden case through add_product (exercising de-duplication)"""
    import sumcheck_amd as sc
    tabs = golden_tables(case)
    if device is not None:
        import torch
        mles = [sc.DenseMultilinearExtension(case["nv"], torch.from_numpy(t.view(np.int64)).to(device)) for t in tabs]
    else:
        mles = [sc.DenseMultilinearExtension(case["nv"], t) for t in tabs]
    coeffs = {tuple(ix): c for c, ix in case["products"]}
    poly = sc.ListOfProductsOfPolynomials(case["nv"])
    for k, shape in enumerate(case["shapes"]):
        c = mont([case["products"][k][0]])[0]
        poly.add_product([mles[i] for i in shape], c)
    return poly, mles


def random_case(rng: np.random.Generator, nv: int, shapes, n_tables: int, seed: int):
    """synthetic (SplitMix64) tables + coefficients -> (cref.PolyDesc builder inputs)"""
    tabs = [cref.synth_table(seed, s, 1 << nv) for s in range(n_tables)]
    coefs = cref.synth_table(seed, 1000, len(shapes))
    return tabs, coefs


def desc_from(nv, shapes, tabs, coefs) -> cref.PolyDesc:
    """flatten like add_product does (first-occurrence order)"""
    order, remap = [], {}
    prods = []
    for k, sh in enumerate(shapes):
        ix = []
        for t in sh:
            if t not in remap:
                remap[t] = len(order)
                order.append(t)
            ix.append(remap[t])
        prods.append((coefs[k], ix))
    return cref.PolyDesc(nv, prods, [tabs[t] for t in order])


def hip_poly_from(nv, shapes, tabs, coefs, device=None):
    import sumcheck_amd as sc
    if device is not None:
        import torch
        mles = [sc.DenseMultilinearExtension(nv, torch.from_numpy(t.view(np.int64)).to(device)) for t in tabs]
    else:
        mles = [sc.DenseMultilinearExtension(nv, t) for t in tabs]
    poly = sc.ListOfProductsOfPolynomials(nv)
    for k, sh in enumerate(shapes):
        poly.add_product([mles[i] for i in sh], coefs[k])
    return poly, mles
