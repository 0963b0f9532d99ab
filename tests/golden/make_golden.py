#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the big-integer oracle (oracle/pyoracle.py).

The reference (Rust) cannot be built or imported in this image and its tests hold no golden
vectors, so these fixtures are produced by the independent Python
restatement; oracle/oracle.c and the HIP library must both reproduce them byte for byte.
Field elements are stored as hex strings of the CANONICAL integer; "mont" entries additionally
store the 4 x u64 Montgomery limbs to pin the boundary conversion.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def hx(x):
    return "%064x" % x


def ml_case(name, nv, shapes, n_tables, seed):
    tabs = [po.synth_table(po.SEED + seed, s, 1 << nv) for s in range(n_tables)]
    coefs = po.synth_table(po.SEED + seed, 1000, len(shapes))
    poly = po.ListOfProductsOfPolynomials(nv)
    for c, sh in zip(coefs, shapes):
        poly.add_product([tabs[i] for i in sh], c)
    # flattened order = first-occurrence order
    flat_ids = [next(i for i, t in enumerate(tabs) if t is ft) for ft in poly.flattened_ml_extensions]
    # (a) interactive run with fixed non-FS challenges
    chal = po.synth_table(po.SEED + seed, 2000, nv)
    st = po.prover_init(poly)
    rounds = []
    v = None
    for i in range(nv):
        rounds.append(po.prove_round(st, v))
        v = chal[i]
    final_tables = st.flattened_ml_extensions
    # (b) Fiat-Shamir run
    proof, st2 = po.ml_prove_as_subprotocol(po.Blake2b512Rng(), poly)
    s = po.extract_sum(proof)
    point, expected = po.ml_verify(poly.info(), s, proof)
    assert poly.evaluate(point) == expected
    assert st2.randomness == point
    # independent sum
    tot = 0
    for c, sh in zip(coefs, shapes):
        for b in range(1 << nv):
            pr = c
            for i in sh:
                pr = pr * tabs[i][b] % po.P
            tot = (tot + pr) % po.P
    assert tot == s
    case = {
        "name": name, "nv": nv, "seed": po.SEED + seed, "shapes": shapes, "n_tables": n_tables,
        "flattened_table_ids": flat_ids,
        "products": [[hx(c), ix] for c, ix in poly.products],
        "tables": [[hx(x) for x in t] for t in tabs],
        "tables_mont0": [list(po.to_mont_limbs(t[0])) for t in tabs],
        "challenges": [hx(x) for x in chal],
        "rounds": [[hx(x) for x in r] for r in rounds],
        "final_tables": [[hx(x) for x in t] for t in final_tables],
        "fs_proof": [[hx(x) for x in r] for r in proof],
        "fs_randomness": [hx(x) for x in st2.randomness],
        "sum": hx(s), "subclaim_expected": hx(expected),
    }
    with open(os.path.join(OUT, f"ml_{name}.json"), "w") as f:
        json.dump(case, f, indent=0)


def gkr_case(dim, seed):
    rnd = random.Random(seed)
    idxs = sorted(rnd.sample(range(1 << (3 * dim)), 1 << dim))
    f1 = {i: rnd.randrange(po.P) for i in idxs}
    f2 = [rnd.randrange(po.P) for _ in range(1 << dim)]
    f3 = [rnd.randrange(po.P) for _ in range(1 << dim)]
    g = [rnd.randrange(po.P) for _ in range(dim)]
    h_g, f1_g = po.initialize_phase_one(f1, 3 * dim, f3, g)
    m1, m2, u, v = po.gkr_prove(po.Blake2b512Rng(), f1, f2, f3, g)
    f1_gu = po.initialize_phase_two(f1_g, dim, u)
    s = 0
    for xy, val in f1_g.items():
        s = (s + val * f2[xy & ((1 << dim) - 1)] * f3[xy >> dim]) % po.P
    assert (m1[0][0] + m1[0][1]) % po.P == s
    uu, vv, exp = po.gkr_verify(po.Blake2b512Rng(), dim, m1, m2, s)
    assert po.gkr_verify_subclaim(f1, f2, f3, g, uu, vv, exp)
    case = {
        "dim": dim, "f1_idx": idxs, "f1_vals": [hx(f1[i]) for i in idxs], "f2": [hx(x) for x in f2], "f3": [hx(x) for x in f3],
        "g": [hx(x) for x in g], "h_g": [hx(x) for x in h_g],
        "f1_g_idx": sorted(f1_g), "f1_g_vals": [hx(f1_g[i]) for i in sorted(f1_g)],
        "f1_gu": [hx(x) for x in f1_gu],
        "phase1": [[hx(x) for x in r] for r in m1], "phase2": [[hx(x) for x in r] for r in m2],
        "u": [hx(x) for x in u], "v": [hx(x) for x in v], "sum": hx(s), "expected": hx(exp),
    }
    with open(os.path.join(OUT, f"gkr_dim{dim}.json"), "w") as f:
        json.dump(case, f, indent=0)


def transcript_case():
    rnd = random.Random(7)
    r = po.Blake2b512Rng()
    ops = []
    for step in range(12):
        if step % 3 != 2:
            m = bytes(rnd.randrange(256) for _ in range(rnd.choice([0, 1, 16, 127, 128, 129, 300])))
            r.feed_bytes(m)
            ops.append(["feed", m.hex()])
        else:
            n = rnd.choice([1, 8, 63, 64, 65, 127, 777])
            ops.append(["fill", n, r.fill_bytes(n).hex()])
        if step % 4 == 3:
            ops.append(["sample", hx(po.sample_fr(r))])
    r2 = po.Blake2b512Rng()
    r2.feed_bytes(po.ser_poly_info(3, 7))
    evs = [rnd.randrange(po.P) for _ in range(4)]
    r2.feed_bytes(po.ser_prover_msg(evs))
    ops2 = {"info": [3, 7], "msg": [hx(e) for e in evs], "sample": hx(po.sample_fr(r2)), "next64": r2.fill_bytes(64).hex()}
    with open(os.path.join(OUT, "transcript.json"), "w") as f:
        json.dump({"ops": ops, "structured": ops2, "blake2b_abc": __import__("hashlib").blake2b(b"abc", digest_size=64).hexdigest()}, f, indent=0)


if __name__ == "__main__":
    ml_case("nv1_trivial", 1, [[0, 1, 2, 3], [4, 5, 6, 7, 8], [0, 0, 4, 4], [9, 10, 11, 12, 9, 10], [3, 2, 1, 0]], 13, 1)
    ml_case("nv2_single", 2, [[0]], 1, 2)
    ml_case("nv3_c1shape", 3, [[0, 1]], 2, 3)
    ml_case("nv6_shared", 6, [[2, 3, 0], [1, 4, 4], [3, 2, 1], [0, 0], [4]], 5, 4)   # reference test.rs:224-252
    ml_case("nv6_c3shape", 6, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10, 5)          # BASELINE C3 product shape
    ml_case("nv7_c2shape", 7, [[0, 1, 2]], 3, 6)                                       # BASELINE C2 product shape
    ml_case("nv5_deg12", 5, [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], [3, 3, 3], [1, 2, 3, 4, 5, 6, 7, 8, 9]], 12, 7)
    ml_case("nv8_bench", 8, [[0, 1, 2], [3, 4, 5]], 6, 8)                              # sumcheck-benches shape
    gkr_case(2, 21)
    gkr_case(4, 22)
    gkr_case(6, 23)
    transcript_case()
    print("golden fixtures written to", OUT)
