"""CPU tests of the oracle itself (no GPU): both restatements against the golden fixtures, against each
other, against published known answers, and against the algebraic relations the reference's tests assert."""
import hashlib

import ctypes as C

import numpy as np
import pytest

from oracle import cref
from oracle import pyoracle as po
from tests import helpers as H


def test_field_constants():
    # recomputed, not trusted
    assert po.P.bit_length() == 255
    assert pow(7, po.P - 1, po.P) == 1
    assert po.R == int("1824b159acc5056f998c4fefecbc4ff55884b7fa0003480200000001fffffffe", 16)
    assert po.R2 == int("0748d9d99f59ff1105d314967254398f2b6cedcb87925c23c999e990f3f29c6d", 16)
    assert (-pow(po.P, -1, 1 << 64)) % (1 << 64) == 0xFFFFFFFEFFFFFFFF
    assert (-pow(po.P, -1, 1 << 32)) % (1 << 32) == 0xFFFFFFFF
    assert po.P % (1 << 32) == 1
    one = cref.ints_to_mont([1])[0]
    assert [int(x) for x in one] == [0x00000001FFFFFFFE, 0x5884B7FA00034802, 0x998C4FEFECBC4FF5, 0x1824B159ACC5056F]


R3_PUBLISHED = int("6e2a5bb9c8db33e973d13c71c7b5f4181b3e0d188cf06990c62c1807439b73af", 16)  # 2^768 mod p: the R3 constant of the published BLS12-381 scalar-field implementations


def lagrange_over_Q(ys, x):
    """sum_i y_i prod_{j != i} (x - j) / (i - j), exactly, with Python fractions -- independent of every modular implementation here"""
    from fractions import Fraction
    tot = Fraction(0)
    for i, y in enumerate(ys):
        t = Fraction(y)
        for j in range(len(ys)):
            if j != i:
                t *= Fraction(x - j, i - j)
        tot += t
    assert tot.denominator == 1
    return int(tot)


def interpolation_known_answers():
    """integer polynomials with SMALL integer coefficients, so that the answer is an integer computed over Q and only then reduced mod p
    (no modular inverse is involved in producing the expectation): lengths on both sides of the reference's tier switches (20, 33
    points: verifier.rs:256-322)"""
    rng = np.random.default_rng(20241008)
    out = []
    for n in (2, 3, 4, 7, 20, 21, 33, 34):
        coef = [int(c) for c in rng.integers(-50, 50, size=n)]
        f = lambda v, coef=coef: sum(c * v ** i for i, c in enumerate(coef))
        ys = [f(i) for i in range(n)]
        for x in (n, n + 5, -3, 1000003):
            want = lagrange_over_Q(ys, x)
            assert want == f(x)
            out.append(([y % po.P for y in ys], x % po.P, want % po.P))
    return out


def test_r3_published_constant_through_the_c_field_multiplier():
    """R^3 mod p by DEPENDENT Montgomery squarings: mont(R)^2 = mont(R^2) whose raw limbs are R^3 (the published literal); two more
    squarings against Python's pow"""
    assert pow(2, 768, po.P) == R3_PUBLISHED
    x = cref.ints_to_mont([po.R])  # Montgomery form of the integer R: raw limbs R^2
    L = cref.lib()
    u64p = C.POINTER(C.c_uint64)
    e = 1
    for step in range(3):
        out = np.empty_like(x)
        L.orc_fr_mul(x[0].ctypes.data_as(u64p), x[0].ctypes.data_as(u64p), out[0].ctypes.data_as(u64p))
        x, e = out, 2 * e
        raw = sum(int(v) << (64 * i) for i, v in enumerate(x[0]))
        assert raw == pow(2, 256 * (e + 1), po.P)
        if step == 0:
            assert raw == R3_PUBLISHED


def test_interpolate_known_answers_over_the_rationals():
    for ys, x, want in interpolation_known_answers():
        assert po.interpolate_uni_poly(ys, x) == want
        assert cref.mont_to_ints(cref.interpolate_uni_poly(cref.ints_to_mont(ys), cref.ints_to_mont([x])[0])) == [want]


def test_field_ops_c_vs_bigint():
    rng = np.random.default_rng(1)
    edge = [0, 1, 2, po.P - 1, po.P - 2, (1 << 255) % po.P, po.R, po.R2]
    vals = edge + [int.from_bytes(rng.bytes(32), "little") % po.P for _ in range(64)]
    for a in vals[:24]:
        for b in vals[:24]:
            A, B = cref.ints_to_mont([a])[0], cref.ints_to_mont([b])[0]
            assert cref.mont_to_ints(cref.fr_binop("mul", A, B)) == [a * b % po.P]
            assert cref.mont_to_ints(cref.fr_binop("add", A, B)) == [(a + b) % po.P]
            assert cref.mont_to_ints(cref.fr_binop("sub", A, B)) == [(a - b) % po.P]


def test_blake2b_known_answer_and_c_impl():
    # RFC 7693 appendix A
    kat = ("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
           "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    assert hashlib.blake2b(b"abc", digest_size=64).hexdigest() == kat
    assert cref.blake2b512(b"abc").hex() == kat
    assert H.load("transcript.json")["blake2b_abc"] == kat
    rng = np.random.default_rng(2)
    for n in [0, 1, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4096]:
        m = rng.bytes(n)
        assert cref.blake2b512(m) == hashlib.blake2b(m, digest_size=64).digest()


# ---- published known answers (third-party pins) ----------------------------------------------
TWO_ADIC_ROOT = 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B  # BLS12-381 Fr: 7^((p-1)/2^32), order 2^32


def _rfc7693_seq(n, seed):  # selftest_seq of RFC 7693 appendix E
    a, b, out = (0xDEAD4BAD * seed) & 0xFFFFFFFF, 1, bytearray()
    for _ in range(n):
        t = (a + b) & 0xFFFFFFFF
        a, b = b, t
        out.append((t >> 24) & 0xFF)
    return bytes(out)


def test_two_adic_root_of_unity_through_the_c_field_multiplier():
    """A 223-bit exponentiation (334 dependent Montgomery products of oracle.c's fr_mul) must land on the published generator
    of the 2^32-th roots of unity of BLS12-381 Fr; 31 more squarings give -1, one more gives 1."""
    assert pow(7, (po.P - 1) >> 32, po.P) == TWO_ADIC_ROOT
    e = (po.P - 1) >> 32
    base = cref.ints_to_mont([7])[0]
    acc = cref.ints_to_mont([1])[0]
    for bit in bin(e)[2:]:
        acc = cref.fr_binop("mul", acc, acc)
        if bit == "1":
            acc = cref.fr_binop("mul", acc, base)
    assert cref.mont_to_ints(acc) == [TWO_ADIC_ROOT]
    for _ in range(31):
        acc = cref.fr_binop("mul", acc, acc)
    assert cref.mont_to_ints(acc) == [po.P - 1]
    assert cref.mont_to_ints(cref.fr_binop("mul", acc, acc)) == [1]


def test_blake2b_rfc7693_selftest_multi_block():
    """RFC 7693 appendix E: the grand hash over inputs of 0, 3, 128, 129, 255 and 1024 bytes (one to eight compression blocks),
    keyed and unkeyed, four digest sizes, is a published constant.  hashlib reproduces it (so hashlib is a faithful BLAKE2b on
    multi-block inputs), and both oracle implementations agree with hashlib on the self-test's own unkeyed 64-byte cases and on a
    168-byte ProverMsg-sized input (8-byte length + 5 x 32 bytes: two blocks, unlike "abc")."""
    grand = hashlib.blake2b(digest_size=32)
    for outlen in (20, 32, 48, 64):
        for inlen in (0, 3, 128, 129, 255, 1024):
            data = _rfc7693_seq(inlen, inlen)
            grand.update(hashlib.blake2b(data, digest_size=outlen).digest())
            grand.update(hashlib.blake2b(data, digest_size=outlen, key=_rfc7693_seq(outlen, outlen)).digest())
    assert grand.hexdigest() == "c23a7800d98123bd10f506c61e29da5603d763b8bbad2e737f5e765a7bccd475"
    for inlen in (0, 3, 128, 129, 168, 232, 255, 1024):
        data = _rfc7693_seq(inlen, inlen)
        want = hashlib.blake2b(data, digest_size=64).digest()
        assert cref.blake2b512(data) == want
        r = po.Blake2b512Rng()
        r.feed_bytes(data)
        assert r.fill_bytes(64) == want  # the first squeeze of the transcript is the plain digest of what was fed (rng.rs:62-63)


def test_transcript_golden_both_oracles():
    g = H.load("transcript.json")
    for mk in (po.Blake2b512Rng, cref.Rng):
        r = mk()
        for op in g["ops"]:
            if op[0] == "feed":
                r.feed_bytes(bytes.fromhex(op[1]))
            elif op[0] == "fill":
                assert r.fill_bytes(op[1]).hex() == op[2]
            else:
                got = po.sample_fr(r) if mk is po.Blake2b512Rng else cref.mont_to_ints(r.sample_fr())[0]
                assert got == H.hx(op[1])
    s = g["structured"]
    r = cref.Rng()
    r.feed_poly_info(*s["info"])
    r.feed_prover_msg(H.mont(s["msg"]))
    assert cref.mont_to_ints(r.sample_fr())[0] == H.hx(s["sample"])
    assert r.fill_bytes(64).hex() == s["next64"]


def test_interpolate_known_answer():
    # the only literal known answer in the reference tree: verifier.rs:327-331
    assert po.interpolate_uni_poly([0, 1, 4, 9], 3) == 9
    got = cref.interpolate_uni_poly(cref.ints_to_mont([0, 1, 4, 9]), cref.ints_to_mont([3])[0])
    assert cref.mont_to_ints(got) == [9]
    # degree-5 polynomial evaluated off the nodes
    rng = np.random.default_rng(3)
    coef = [int.from_bytes(rng.bytes(32), "little") % po.P for _ in range(6)]
    f = lambda x: sum(c * pow(x, i, po.P) for i, c in enumerate(coef)) % po.P
    ys = [f(i) for i in range(6)]
    x = int.from_bytes(rng.bytes(32), "little") % po.P
    assert po.interpolate_uni_poly(ys, x) == f(x)
    assert cref.mont_to_ints(cref.interpolate_uni_poly(cref.ints_to_mont(ys), cref.ints_to_mont([x])[0])) == [f(x)]


@pytest.mark.parametrize("name", H.ml_cases())
def test_c_oracle_matches_golden_rounds(name):
    case = H.load(name)
    d = H.oracle_desc(case)
    assert [int(x) for x in H.golden_tables(case)[0][0]] == case["tables_mont0"][0]
    for threads in (1, 4):
        p = cref.Prover(d, threads=threads)
        chal = H.mont(case["challenges"])
        v = None
        for i in range(case["nv"]):
            got = p.prove_round(v)
            assert cref.mont_to_ints(got) == [H.hx(x) for x in case["rounds"][i]], (name, i)
            v = chal[i]
        _, tabs, rnd = p.state()
        assert rnd == case["nv"]
        for u in range(len(d.tables)):
            assert cref.mont_to_ints(tabs[u]) == [H.hx(x) for x in case["final_tables"][u]]


@pytest.mark.parametrize("name", H.ml_cases())
def test_c_oracle_fs_proof_and_verifier(name):
    case = H.load(name)
    d = H.oracle_desc(case)
    proof, rand = cref.ml_prove(d)
    for i in range(case["nv"]):
        assert cref.mont_to_ints(proof[i]) == [H.hx(x) for x in case["fs_proof"][i]]
    assert cref.mont_to_ints(rand) == [H.hx(x) for x in case["fs_randomness"]]
    s = H.mont([case["sum"]])[0]
    ok, point, exp = cref.ml_verify((d.max_multiplicands, d.num_vars), s, proof)
    assert ok
    assert cref.mont_to_ints(exp) == [H.hx(case["subclaim_expected"])]
    assert np.array_equal(point, rand)  # test.rs:119
    assert np.array_equal(cref.poly_evaluate(d, point), exp)  # test.rs:71-74
    bad = s.copy()
    bad[0] ^= np.uint64(1)
    ok2, _, _ = cref.ml_verify((d.max_multiplicands, d.num_vars), bad, proof)
    assert not ok2


def test_prover_state_machine_errors():
    case = H.load("ml_nv3_c1shape.json")
    d = H.oracle_desc(case)
    p = cref.Prover(d)
    r = H.mont(case["challenges"])[0]
    with pytest.raises(RuntimeError, match="first round should be prover first"):
        p.prove_round(r)
    p.prove_round(None)
    with pytest.raises(RuntimeError, match="verifier message is empty"):
        p.prove_round(None)
    p.prove_round(r)
    p.prove_round(r)
    with pytest.raises(RuntimeError, match="Prover is not active"):
        p.prove_round(r)
    with pytest.raises(RuntimeError, match="Attempt to prove a constant"):
        cref.Prover(cref.PolyDesc(0, [(r, [0])], [np.zeros((1, 4), np.uint64)]))


def test_pyoracle_vs_c_random_shapes():
    rng = np.random.default_rng(11)
    for trial in range(4):
        nv = int(rng.integers(1, 6))
        nt = int(rng.integers(1, 6))
        shapes = [[int(x) for x in rng.integers(0, nt, size=int(rng.integers(1, 5)))] for _ in range(int(rng.integers(1, 4)))]
        tabs = [cref.synth_table(100 + trial, s, 1 << nv) for s in range(nt)]
        coefs = cref.synth_table(100 + trial, 1000, len(shapes))
        d = H.desc_from(nv, shapes, tabs, coefs)
        itabs = [cref.mont_to_ints(t) for t in tabs]
        poly = po.ListOfProductsOfPolynomials(nv)
        for k, sh in enumerate(shapes):
            poly.add_product([itabs[i] for i in sh], cref.mont_to_ints(coefs[k])[0])
        proof_py = po.ml_prove(poly)
        proof_c, _ = cref.ml_prove(d, threads=3)
        for i in range(nv):
            assert cref.mont_to_ints(proof_c[i]) == proof_py[i]


def test_synth_generator_c_vs_py():
    for stream in (0, 7):
        a = cref.synth_table(po.SEED, stream, 33, first=5)
        for i in range(33):
            assert tuple(int(x) for x in a[i]) == po.synth_mont_limbs(po.SEED, stream, 5 + i)


@pytest.mark.parametrize("name", H.gkr_cases())
def test_gkr_c_oracle_matches_golden(name):
    g = H.load(name)
    dim = g["dim"]
    idx = np.asarray(g["f1_idx"], dtype=np.uint64)
    vals, f2, f3, gg = H.mont(g["f1_vals"]), H.mont(g["f2"]), H.mont(g["f3"]), H.mont(g["g"])
    h_g, gi, gv = cref.gkr_phase_one(idx, vals, dim, f3, gg)
    assert cref.mont_to_ints(h_g) == [H.hx(x) for x in g["h_g"]]
    assert [int(x) for x in gi] == g["f1_g_idx"]
    assert cref.mont_to_ints(gv) == [H.hx(x) for x in g["f1_g_vals"]]
    f1_gu = cref.gkr_phase_two(gi, gv, dim, H.mont(g["u"]))
    assert cref.mont_to_ints(f1_gu) == [H.hx(x) for x in g["f1_gu"]]
    proof, uv = cref.gkr_prove(idx, vals, dim, f2, f3, gg, threads=2)
    for i in range(dim):
        assert cref.mont_to_ints(proof[0, i]) == [H.hx(x) for x in g["phase1"][i]]
        assert cref.mont_to_ints(proof[1, i]) == [H.hx(x) for x in g["phase2"][i]]
    assert cref.mont_to_ints(uv[0]) == [H.hx(x) for x in g["u"]]
    assert cref.mont_to_ints(uv[1]) == [H.hx(x) for x in g["v"]]


def test_gkr_naive_sum_and_subclaim_pyoracle():
    # reference gkr_round_sumcheck/test.rs:24-45,71-88 at dim 3 with the O(4^dim) naive sum
    import random
    rnd = random.Random(9)
    dim = 3
    f1 = {i: rnd.randrange(po.P) for i in rnd.sample(range(1 << (3 * dim)), 1 << dim)}
    f2 = [rnd.randrange(po.P) for _ in range(1 << dim)]
    f3 = [rnd.randrange(po.P) for _ in range(1 << dim)]
    g = [rnd.randrange(po.P) for _ in range(dim)]
    f1_g, _ = po.sparse_fix_variables(f1, 3 * dim, g)
    naive = 0
    for x in range(1 << dim):
        xs = [(x >> k) & 1 for k in range(dim)]
        f1_gx, nvr = po.sparse_fix_variables(f1_g, 2 * dim, xs)
        dense = po.sparse_to_dense(f1_gx, nvr)
        for y in range(1 << dim):
            naive = (naive + dense[y] * f2[x] * f3[y]) % po.P
    m1, m2, u, v = po.gkr_prove(po.Blake2b512Rng(), f1, f2, f3, g)
    assert (m1[0][0] + m1[0][1]) % po.P == naive  # test_extract
    uu, vv, exp = po.gkr_verify(po.Blake2b512Rng(), dim, m1, m2, naive)
    assert po.gkr_verify_subclaim(f1, f2, f3, g, uu, vv, exp)  # test_small


def test_pin_reads_every_fixture_field():
    """rust-shim/tests/dump_vectors.rs is the one-command pin against real arkworks (it cannot be built here: no cargo).  It must stay in
    step with tests/golden/: every field make_golden.py writes is read by the Rust file (as `["field"]`), and every fixture file is
    loaded by it -- so the day cargo exists nothing the fixtures hold goes uncompared.  `seed` only documents how the tables were
    synthesised (the Rust side reads the tables themselves)."""
    import glob
    import json
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rs = open(os.path.join(root, "rust-shim", "tests", "dump_vectors.rs")).read()
    read = set(re.findall(r'\["([A-Za-z0-9_]+)"\]', rs))
    metadata_only = {"seed"}
    files = sorted(glob.glob(os.path.join(root, "tests", "golden", "*.json")))
    assert len(files) >= 12
    for f in files:
        stem = os.path.basename(f)[:-5]
        if stem.startswith("ml_"):
            assert f'"{stem[3:]}"' in rs, f"{stem}: not in ml_proof_vectors' / ml_interactive_vectors' case lists"
        elif stem.startswith("gkr_dim"):
            assert re.search(r"for dim in \[[^\]]*(?<![0-9])%s(?![0-9])" % stem[7:], rs), f"{stem}: not in gkr_vectors' dim list"
        else:
            assert f'load("{stem}")' in rs, f"{stem}: never loaded"
        d = json.load(open(f))
        keys = set(d)
        if stem == "transcript":
            keys |= set(d["structured"])
        missing = sorted(k for k in keys - metadata_only if k not in read)
        assert not missing, f"{stem}.json: fields the Rust pin never reads: {missing}"
    # ... and the generator writes nothing else: the keys of its `case = {...}` literals are the keys of the fixtures
    gen = open(os.path.join(root, "tests", "golden", "make_golden.py")).read()
    gen_keys = set(re.findall(r'"([A-Za-z0-9_]+)": ', gen))  # ("key": value -- not `if __name__ == "__main__":`)
    assert gen_keys - metadata_only <= read, sorted(gen_keys - metadata_only - read)
