"""GPU parity tests (-m gpu): the HIP path through the C ABI against the golden fixtures and the C oracle,
bit-exact on identical inputs; plus size-independent properties at BASELINE's full sizes."""
import ctypes as C
import os

import numpy as np
import pytest

import sumcheck_amd as sc
from oracle import cref
from sumcheck_amd import _lib, field
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert sc.lib().sc_device_count() > 0, "these tests need a HIP device (no fallback path exists)"


def _torch_dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("op,name", [(0, "mul"), (1, "add"), (2, "sub"), (3, "mul"), (4, "mul"), (5, "mulu")])
def test_field_arithmetic_every_implementation(op, name):
    """each device implementation of Fr mul/add/sub against the oracle on edge values and 100k random pairs"""
    from oracle import pyoracle as po
    rng = np.random.default_rng(op)
    edge = [0, 1, 2, po.P - 1, po.P - 2, (po.P - 1) // 2, (po.P + 1) // 2, po.R, po.R2, (1 << 255) % po.P, 0xFFFFFFFF, 1 << 32,
            (1 << 64) - 1, 1 << 64, po.P - (1 << 32), po.P - (1 << 224)]
    ea = cref.ints_to_mont([x for x in edge for _ in edge])
    eb = cref.ints_to_mont([y for _ in edge for y in edge])
    n = 100_000
    a = np.concatenate([ea, cref.synth_table(1, 2 * op, n)])
    b = np.concatenate([eb, cref.synth_table(1, 2 * op + 1, n)])
    if name == "mulu":
        b = np.repeat(b[len(edge) * 3 + 5][None, :], a.shape[0], axis=0)  # uniform operand
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    out = np.empty_like(a)
    _lib.check(sc.lib().sc_fr_elementwise(op, C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(out.ctypes.data), a.shape[0]))
    # oracle: canonical residues via python ints on a sample + the C oracle on everything
    want = np.empty_like(a)
    fn = {"mul": "mul", "mulu": "mul", "add": "add", "sub": "sub"}[name]
    L = cref.lib()
    f = getattr(L, "orc_fr_" + fn)
    u64p = C.POINTER(C.c_uint64)
    for i in range(a.shape[0]):
        f(a[i].ctypes.data_as(u64p), b[i].ctypes.data_as(u64p), want[i].ctypes.data_as(u64p))
    assert np.array_equal(out, want)
    ai, bi, oi = cref.mont_to_ints(a[:300]), cref.mont_to_ints(b[:300]), cref.mont_to_ints(out[:300])
    pyf = {"mul": lambda x, y: x * y % po.P, "add": lambda x, y: (x + y) % po.P, "sub": lambda x, y: (x - y) % po.P}[fn]
    assert oi == [pyf(x, y) for x, y in zip(ai, bi)]


@pytest.mark.parametrize("op", [0, 3, 4])
def test_two_adic_root_of_unity_through_the_device_multipliers(op):
    """Published known answer (third-party pin): 7^((p-1)/2^32) is the generator of BLS12-381 Fr's 2^32-th roots of unity,
    0x16a2...0d2b.  334 DEPENDENT Montgomery products through the device multiplier (op 0: the production carry-free fe_mul;
    3 / 4: the saturated CIOS and Comba products) must land on it; 31 more squarings give -1, one more 1."""
    from oracle import pyoracle as po
    from tests.test_oracle import TWO_ADIC_ROOT

    def dmul(x, y):
        out = np.empty_like(x)
        _lib.check(sc.lib().sc_fr_elementwise(op, C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_void_p(out.ctypes.data), x.shape[0]))
        return out

    base = np.ascontiguousarray(cref.ints_to_mont([7, 7, 7]))
    acc = np.ascontiguousarray(cref.ints_to_mont([1, 1, 1]))
    for bit in bin((po.P - 1) >> 32)[2:]:
        acc = dmul(acc, acc)
        if bit == "1":
            acc = dmul(acc, base)
    assert cref.mont_to_ints(acc) == [TWO_ADIC_ROOT] * 3
    for _ in range(31):
        acc = dmul(acc, acc)
    assert cref.mont_to_ints(acc) == [po.P - 1] * 3
    assert cref.mont_to_ints(dmul(acc, acc)) == [1] * 3


@pytest.mark.parametrize("op", [0, 3, 4])
def test_r3_published_constant_through_the_device_multipliers(op):
    """Published known answer: R^3 = 2^768 mod p = 0x6e2a...73af.  mont(R) squared is mont(R^2), whose raw limbs ARE R^3 -- for the
    production multiplier (op 0) this goes through the 9 x 29-bit representation and its 2^261 radix, so the literal pins the 2^5
    compensation; two more dependent squarings against Python's pow."""
    from oracle import pyoracle as po
    from tests.test_oracle import R3_PUBLISHED
    x = np.ascontiguousarray(cref.ints_to_mont([po.R] * 5))
    e = 1
    for step in range(3):
        out = np.empty_like(x)
        _lib.check(sc.lib().sc_fr_elementwise(op, C.c_void_p(x.ctypes.data), C.c_void_p(x.ctypes.data), C.c_void_p(out.ctypes.data), x.shape[0]))
        x, e = out, 2 * e
        raws = [sum(int(v) << (64 * i) for i, v in enumerate(row)) for row in x]
        assert raws == [pow(2, 256 * (e + 1), po.P)] * 5
        if step == 0:
            assert raws[0] == R3_PUBLISHED


def test_add_sub_mul_edge_identities_on_the_device():
    """literal identities around (p - 1) / 2 and -1 through every device add / sub / mul: the places where a lazy or carry-free
    representation has to wrap exactly"""
    from oracle import pyoracle as po
    h, m1 = (po.P - 1) // 2, po.P - 1
    cases = [  # (op name, a, b, expected)
        ("add", h, h, m1), ("add", h, h + 1, 0), ("add", m1, 1, 0), ("add", m1, m1, po.P - 2), ("add", h + 1, h + 1, 1),
        ("sub", 0, 1, m1), ("sub", h, h + 1, m1), ("sub", 0, m1, 1), ("sub", m1, m1, 0), ("sub", 1, m1, 2), ("sub", h, m1, h + 1),
        ("mul", m1, m1, 1), ("mul", 2, h + 1, 1), ("mul", m1, h, h + 1), ("mul", m1, 1, m1), ("mul", h + 1, h + 1, pow(4, -1, po.P)),
    ]
    for ops, name in (((1,), "add"), ((2,), "sub"), ((0, 3, 4), "mul")):
        sel = [c for c in cases if c[0] == name]
        a = np.ascontiguousarray(cref.ints_to_mont([c[1] for c in sel]))
        b = np.ascontiguousarray(cref.ints_to_mont([c[2] for c in sel]))
        for op in ops:
            out = np.empty_like(a)
            _lib.check(sc.lib().sc_fr_elementwise(op, C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(out.ctypes.data), a.shape[0]))
            assert cref.mont_to_ints(out) == [c[3] for c in sel], (name, op)


def _interactive(poly, challenges, borrow=False):
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=borrow)
    msgs, v = [], None
    for i in range(poly.num_variables):
        msgs.append(sc.IPForMLSumcheck.prove_round(st, v).evaluations)
        v = sc.VerifierMsg(challenges[i])
    return st, msgs


@pytest.mark.parametrize("name", H.ml_cases())
@pytest.mark.parametrize("where", ["host", "device", "borrow"])
def test_golden_rounds(name, where):
    case = H.load(name)
    dev = None if where == "host" else _torch_dev()
    poly, mles = H.hip_poly(case, device=dev)
    assert len(poly.flattened_ml_extensions) == len(case["flattened_table_ids"])
    st, msgs = _interactive(poly, H.mont(case["challenges"]), borrow=(where == "borrow"))
    for i in range(case["nv"]):
        assert field.to_ints(msgs[i]) == [H.hx(x) for x in case["rounds"][i]], (name, i)
    assert st.round == case["nv"]
    finals = st.flattened_ml_extensions
    for u, t in enumerate(finals):
        assert t.num_vars == 1
        assert field.to_ints(t.evaluations) == [H.hx(x) for x in case["final_tables"][u]]
    assert field.to_ints(st.randomness) == [H.hx(x) for x in case["challenges"][: case["nv"] - 1]]
    if where == "borrow":  # borrowed inputs are never written
        for m, t in zip(mles, H.golden_tables(case)):
            assert np.array_equal(m.evaluations.cpu().numpy().view(np.uint64), t)


@pytest.mark.parametrize("name", H.ml_cases())
def test_golden_fiat_shamir_proof(name):
    case = H.load(name)
    poly, _ = H.hip_poly(case)
    proof = sc.MLSumcheck.prove(poly)
    for i in range(case["nv"]):
        assert field.to_ints(proof[i].evaluations) == [H.hx(x) for x in case["fs_proof"][i]]
    assert field.to_int(sc.MLSumcheck.extract_sum(proof)) == H.hx(case["sum"])  # test.rs:206-213
    # subprotocol flow with pre-fed transcripts (test.rs:99-120)
    pr, vr = sc.Blake2b512Rng.setup(), sc.Blake2b512Rng.setup()
    pr.feed(b"Test Trivial Works"); vr.feed(b"Test Trivial Works")
    proof2, state = sc.MLSumcheck.prove_as_subprotocol(pr, poly)
    sub = sc.MLSumcheck.verify_as_subprotocol(vr, poly.info(), H.mont([case["sum"]])[0], proof2)
    assert np.array_equal(state.randomness, sub.point)  # test.rs:119
    assert np.array_equal(poly.evaluate(sub.point), sub.expected_evaluation)  # test.rs:115-118


SHAPES = [
    (1, 13, [[0, 1, 2, 3], [4, 5, 6, 7, 8, 9, 10, 11, 12, 0, 1, 2], [5, 5, 5, 5, 5], [7, 8, 9, 10, 11, 12], [1, 2, 3, 4, 5, 6, 7]]),  # test_trivial_polynomial
    (9, 5, [[2, 3, 0], [1, 4, 4], [3, 2, 1], [0, 0], [4]]),                      # test_shared_reference
    (12, 8, [[0, 1, 2, 3], [4, 5, 6, 7, 0], [1, 2, 3, 4, 5, 6], [7, 6, 5, 4, 3, 2, 1], [0, 2, 4, 6, 1, 3, 5, 7]]),  # test_normal_polynomial
    (12, 2, [[0, 1]]),                                                           # BASELINE config 1
    (14, 10, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]),                            # BASELINE config 3 shape
    (13, 9, [[0, 1, 2, 3, 4, 5, 6, 7, 8], [0, 0, 0, 0, 0, 0, 0, 0, 0, 0]]),      # generic path (>8 multiplicands)
    (17, 3, [[0, 1, 2]]),                                                        # several grid-stride iterations
    (3, 2, [[0], [1], [0, 1]]),
    (9, 36, [[3 * i, 3 * i + 1, 3 * i + 2] for i in range(11)] + [[33, 34], [35]]),  # > 32 tables: no small-round kernels
    # several big binding rounds in a row with every product length 1..4 and repeated tables: node 1 of those rounds comes from the
    # previous round's sums (claim identity), compared with the oracle message by message
    (16, 7, [[0, 1, 2, 3], [4, 5, 6], [1, 1], [2], [3, 3, 3], [5, 6, 6, 0]]),
    (15, 2, [[0], [1]]),
    (18, 4, [[0, 1], [2, 3], [0, 3]]),
    # launch plans the shapes above do not reach (tests/test_zz_plan_coverage.py):
    (17, 36, [[3 * i, 3 * i + 1, 3 * i + 2] for i in range(12)]),  # more than 32 tables in ONE merged launch: the bound tables stay in the reference layout
    (10, 3, [[0, 1]] * 9 + [[1, 2]] * 9),                          # more products than a launch's arguments describe: the one-block finalize
    (9, 4, [[0, 1, 2, 3]] * 22 + [[1, 2, 3]] * 22),                # node sums beyond the finalize step's LDS
]


@pytest.mark.parametrize("nv,nt,shapes", SHAPES)
def test_random_shapes_vs_oracle(nv, nt, shapes):
    tabs = [cref.synth_table(777 + nv, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(777 + nv, 1000, len(shapes))
    chal = cref.synth_table(777 + nv, 2000, nv)
    d = H.desc_from(nv, shapes, tabs, coefs)
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs)
    op = cref.Prover(d, threads=cref.max_threads())
    st = sc.IPForMLSumcheck.prover_init(poly)
    v = None
    for i in range(nv):
        want = op.prove_round(None if v is None else v.randomness)
        got = sc.IPForMLSumcheck.prove_round(st, v).evaluations
        assert np.array_equal(got, want), f"round {i + 1}"
        v = sc.VerifierMsg(chal[i])
    _, otabs, _ = op.state()
    for u, t in enumerate(st.flattened_ml_extensions):
        assert np.array_equal(t.evaluations, otabs[u])
    # whole Fiat-Shamir proofs agree too
    proof = sc.MLSumcheck.prove(poly)
    want, _ = cref.ml_prove(d, threads=cref.max_threads())
    assert np.array_equal(np.stack([m.evaluations for m in proof]), want)



def test_claim_identity_rounds_survive_state_export_timing_stream_switch_and_reset():
    """Node 1 of a big binding round comes from the previous round's sums kept on the device.  Everything a caller may do between
    two rounds -- read the state, switch timing on, move the handle to another stream, rewind it mid-proof -- must leave every message equal to the
    oracle's."""
    import torch
    nv, shapes, nt = 16, [[0, 1, 2, 3], [4, 5, 6], [1, 1], [2]], 7
    poly, mles, coefs = _device_poly(nv, shapes, nt, 4242)
    tabs = [m.evaluations.cpu().numpy().view(np.uint64) for m in mles]
    d = H.desc_from(nv, shapes, tabs, coefs)
    chal = cref.synth_table(4242, 2000, nv)
    op = cref.Prover(d, threads=cref.max_threads())
    want = []
    v = None
    for i in range(nv):
        want.append(op.prove_round(None if v is None else v))
        v = chal[i]
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    side = torch.cuda.Stream()

    def run(upto, meddle):
        v = None
        for i in range(upto):
            got = sc.IPForMLSumcheck.prove_round(st, v).evaluations
            assert np.array_equal(got, want[i]), f"round {i + 1}"
            v = sc.VerifierMsg(chal[i])
            if meddle and i == 0:
                assert st.round == 1 and len(st.randomness) == 0
            if meddle and i == 1:
                st.set_timing(True)
                t = st.flattened_ml_extensions  # exports (and converts) the bound tables
                assert t[0].evaluations.shape[0] == 1 << (nv - 1)
            if meddle and i == 2:
                _lib.check(sc.lib().sc_prover_set_stream(st._h, C.c_void_p(side.cuda_stream), 0))
            if meddle and i == 4:
                _lib.check(sc.lib().sc_prover_set_stream(st._h, None, 1))
                st.set_timing(False)

    run(nv, meddle=True)
    st.reset()
    run(4, meddle=False)  # abandoned mid-proof, in the middle of the big binding rounds ...
    st.reset()
    run(nv, meddle=False)  # ... and the next proof starts from nothing
    st.reset()
    proof = st.prove(sc.Blake2b512Rng.setup())
    wantp, _ = cref.ml_prove(d, threads=cref.max_threads())
    assert np.array_equal(proof, wantp)


def test_state_machine_errors():
    case = H.load("ml_nv3_c1shape.json")
    poly, _ = H.hip_poly(case)
    r = sc.VerifierMsg(H.mont(case["challenges"])[0])
    st = sc.IPForMLSumcheck.prover_init(poly)
    with pytest.raises(sc.SumcheckError, match="first round should be prover first"):
        sc.IPForMLSumcheck.prove_round(st, r)
    sc.IPForMLSumcheck.prove_round(st, None)
    with pytest.raises(sc.SumcheckError, match="verifier message is empty"):
        sc.IPForMLSumcheck.prove_round(st, None)
    sc.IPForMLSumcheck.prove_round(st, r)
    sc.IPForMLSumcheck.prove_round(st, r)
    with pytest.raises(sc.SumcheckError, match="Prover is not active"):
        sc.IPForMLSumcheck.prove_round(st, r)
    assert st.round == 3
    bad = np.array([0xFFFFFFFFFFFFFFFF] * 4, dtype=np.uint64)  # not a canonical element
    st2 = sc.IPForMLSumcheck.prover_init(poly)
    sc.IPForMLSumcheck.prove_round(st2, None)
    with pytest.raises(sc.SumcheckError, match="canonical"):
        sc.IPForMLSumcheck.prove_round(st2, sc.VerifierMsg(bad))


@pytest.mark.parametrize("nv,k", [(1, 1), (5, 0), (5, 2), (10, 10), (16, 3), (18, 18)])
def test_fix_variables_vs_oracle(nv, k):
    t = cref.synth_table(31, nv, 1 << nv)
    pt = cref.synth_table(31, 99, max(k, 1))[:k]
    want = cref.fix_variables(t, pt)
    mle = sc.DenseMultilinearExtension(nv, t)
    got = mle.fix_variables(pt).evaluations
    assert np.array_equal(got, want)
    import torch
    dm = sc.DenseMultilinearExtension(nv, torch.from_numpy(t.view(np.int64)).to(_torch_dev()))
    got_d = dm.fix_variables(pt).evaluations.cpu().numpy().view(np.uint64)
    assert np.array_equal(got_d, want)


def test_synth_table_device_matches_oracle():
    import torch
    n = 5000
    out = torch.empty((n, 4), dtype=torch.int64, device=_torch_dev())
    _lib.check(sc.lib().sc_synth_table_device(0x5C20241008, 3, 17, n, C.c_void_p(out.data_ptr())))
    assert np.array_equal(out.cpu().numpy().view(np.uint64), cref.synth_table(0x5C20241008, 3, n, first=17))


def _device_poly(nv, shapes, nt, seed):
    import torch
    dev = _torch_dev()
    mles = []
    for s in range(nt):
        t = torch.empty((1 << nv, 4), dtype=torch.int64, device=dev)
        _lib.check(sc.lib().sc_synth_table_device(seed, s, 0, 1 << nv, C.c_void_p(t.data_ptr())))
        mles.append(sc.DenseMultilinearExtension(nv, t))
    coefs = cref.synth_table(seed, 1000, len(shapes))
    poly = sc.ListOfProductsOfPolynomials(nv)
    for k, sh in enumerate(shapes):
        poly.add_product([mles[i] for i in sh], coefs[k])
    return poly, mles, coefs


def test_config2_nv20_bit_exact_vs_oracle():
    """BASELINE config 2: 1 product of 3 multilinears, nv=20, bit-exact vs the CPU oracle."""
    nv, shapes = 20, [[0, 1, 2]]
    poly, mles, coefs = _device_poly(nv, shapes, 3, 0x5C20241008)
    tabs = [m.evaluations.cpu().numpy().view(np.uint64) for m in mles]
    d = H.desc_from(nv, shapes, tabs, coefs)
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly, borrow=True)
    assert np.array_equal(np.stack([m.evaluations for m in proof]), want)
    assert np.array_equal(state.randomness, wrand)


@pytest.mark.parametrize("nv,shapes,nt", [
    (24, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10),      # BASELINE config 3 (headline), 5 GiB of tables
    (24, [[2, 3, 0, 1], [1, 4, 4], [3, 2, 1], [0, 0]], 5),  # C3s: shared tables (shape of reference test.rs:224-252)
])
def test_full_size_properties(nv, shapes, nt):
    """At full size the oracle is too slow; use the size-independent relations the reference's own tests assert:
    verifier acceptance of every round (P_i(0)+P_i(1) == P_{i-1}(r_{i-1})) and the final oracle query
    poly.evaluate(point) == expected_evaluation (test.rs:71-74), the latter computed by the independent
    evaluate-at-a-point kernel (sc_poly_evaluate); plus round 1 against the oracle on a 2^16-point slice via linearity of the sum."""
    poly, mles, coefs = _device_poly(nv, shapes, nt, 0x5C20241008)
    proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly, borrow=True)
    s = sc.MLSumcheck.extract_sum(proof)
    sub = sc.MLSumcheck.verify(poly.info(), s, proof)
    assert np.array_equal(state.randomness, sub.point)
    assert np.array_equal(poly.evaluate(sub.point), sub.expected_evaluation)
    # the final 2-entry tables equal the tables bound at the first nv-1 challenges
    pt = sub.point[: nv - 1]
    for u, t in enumerate(state.flattened_ml_extensions):
        want = poly.flattened_ml_extensions[u].fix_variables(pt).evaluations.cpu().numpy().view(np.uint64)
        assert np.array_equal(t.evaluations, want)


def test_sharded_partial_rounds_single_gpu():
    """SURVEY 8e on one device: G logical shards (contiguous high-bit blocks), per-round integer lane sum +
    sc_wide_reduce, bind_final + gather for the last log2(G) rounds; must equal the unsharded oracle."""
    import torch
    from sumcheck_amd import sharded
    dev = _torch_dev()
    nv, shapes, nt, G = 12, [[0, 1, 2], [1, 3]], 4, 4
    tabs = [cref.synth_table(55, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(55, 1000, len(shapes))
    d = H.desc_from(nv, shapes, tabs, coefs)
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    got, rand = sharded.prove_logical_shards(nv, shapes, tabs, coefs, G, dev)
    assert np.array_equal(got, want)
    assert np.array_equal(rand, wrand)


def test_sharded_tail_prover_is_reused_across_proofs():
    """bench.py --gpus N proves repeatedly: the log2(G)-variable tail prover is built once (sharded.TailEngines) and reloaded
    with every proof's gathered tables.  Two proofs over different tables through the same tail must both match the oracle."""
    from sumcheck_amd import sharded
    dev = _torch_dev()
    nv, shapes, nt, G = 11, [[0, 1, 2, 3], [2], [1, 1]], 4, 8
    coefs = cref.synth_table(56, 1000, len(shapes))
    tail = sharded.TailEngines(shapes, coefs, dev)
    n_loc = 1 << (nv - 3)
    for seed in (56, 57, 58):
        tabs = [cref.synth_table(seed, s, 1 << nv) for s in range(nt)]
        d = H.desc_from(nv, shapes, tabs, coefs)
        want, wrand = cref.ml_prove(d, threads=cref.max_threads())
        engines = [sharded.HipShardEngine(nv - 3, shapes, coefs, [t[g * n_loc:(g + 1) * n_loc] for t in tabs], dev, borrow=True) for g in range(G)]
        got, rand = sharded.prove_sharded(engines, sharded.DistComm(), nv, max(len(sh) for sh in shapes), tail)
        assert np.array_equal(got, want)
        assert np.array_equal(rand, wrand)


# ------------------------------------------------------------------------------------------------------------------
# GKR round sumcheck (BASELINE config 5): reference src/gkr_round_sumcheck/{mod.rs,test.rs}
# ------------------------------------------------------------------------------------------------------------------
def _gkr_inputs_from_golden(g):
    dim = g["dim"]
    f1 = sc.SparseMultilinearExtension(3 * dim, np.asarray(g["f1_idx"], dtype=np.uint64), H.mont(g["f1_vals"]))
    f2 = sc.DenseMultilinearExtension(dim, H.mont(g["f2"]))
    f3 = sc.DenseMultilinearExtension(dim, H.mont(g["f3"]))
    return dim, f1, f2, f3, H.mont(g["g"])


@pytest.mark.parametrize("name", H.gkr_cases())
def test_gkr_golden(name):
    g = H.load(name)
    dim, f1, f2, f3, gg = _gkr_inputs_from_golden(g)
    # shuffled input order must not matter (the reference's f1 is a BTreeMap)
    perm = np.random.default_rng(1).permutation(f1.indices.shape[0])
    f1s = sc.SparseMultilinearExtension(3 * dim, f1.indices[perm], f1.values[perm])
    h_g, f1_g = sc.initialize_phase_one(f1s, f3, gg)
    assert field.to_ints(h_g.evaluations) == [H.hx(x) for x in g["h_g"]]
    assert f1_g.indices.tolist() == g["f1_g_idx"]
    assert field.to_ints(f1_g.values) == [H.hx(x) for x in g["f1_g_vals"]]
    f1_gu = sc.initialize_phase_two(f1_g, H.mont(g["u"]))
    assert field.to_ints(f1_gu.evaluations) == [H.hx(x) for x in g["f1_gu"]]
    proof = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1s, f2, f3, gg)
    for i in range(dim):
        assert field.to_ints(proof.phase1_sumcheck_msgs[i].evaluations) == [H.hx(x) for x in g["phase1"][i]]
        assert field.to_ints(proof.phase2_sumcheck_msgs[i].evaluations) == [H.hx(x) for x in g["phase2"][i]]
    assert field.to_int(proof.extract_sum()) == H.hx(g["sum"])  # gkr test.rs:76-88
    sub = sc.GKRRoundSumcheck.verify(sc.Blake2b512Rng.setup(), dim, proof, H.mont([g["sum"]])[0])
    assert field.to_ints(sub.u) == [H.hx(x) for x in g["u"]] and field.to_ints(sub.v) == [H.hx(x) for x in g["v"]]
    assert field.to_int(sub.expected_evaluation) == H.hx(g["expected"])
    assert sub.verify_subclaim(f1, f2, f3, gg)  # gkr test.rs:58-68
    # the interactive pieces (start_phase1_sumcheck) agree with the fused driver
    st = sc.start_phase1_sumcheck(h_g, f2)
    assert np.array_equal(sc.IPForMLSumcheck.prove_round(st, None).evaluations, proof.phase1_sumcheck_msgs[0].evaluations)


def _random_gkr(dim, seed, collide=False):
    rng = np.random.default_rng(seed)
    n = 1 << dim
    if collide:  # force many (x,y) collisions after binding z, and many x collisions in the scatter
        idx = np.unique((rng.integers(0, 1 << dim, size=4 * n, dtype=np.uint64)) | (rng.integers(0, 4, size=4 * n, dtype=np.uint64) << np.uint64(dim))
                        | (rng.integers(0, 3, size=4 * n, dtype=np.uint64) << np.uint64(2 * dim)))[:n]
    else:
        idx = np.unique(rng.integers(0, 1 << (3 * dim), size=2 * n, dtype=np.uint64))[:n]
    vals = cref.synth_table(seed, 1, idx.shape[0])
    return idx, vals, cref.synth_table(seed, 2, n), cref.synth_table(seed, 3, n), cref.synth_table(seed, 4, dim)


@pytest.mark.parametrize("dim,collide", [(1, False), (3, True), (9, False), (12, True), (16, False)])
def test_gkr_vs_oracle(dim, collide):
    """reference gkr test.rs:71-74 (dim 9) plus collision-heavy inputs; every stage against the C oracle"""
    idx, vals, f2, f3, g = _random_gkr(dim, 1000 + dim, collide)
    f1 = sc.SparseMultilinearExtension(3 * dim, idx, vals)
    mf2, mf3 = sc.DenseMultilinearExtension(dim, f2), sc.DenseMultilinearExtension(dim, f3)
    h_g, f1_g = sc.initialize_phase_one(f1, mf3, g)
    wh, wi, wv = cref.gkr_phase_one(idx, vals, dim, f3, g)
    assert np.array_equal(h_g.evaluations, wh) and np.array_equal(f1_g.indices, wi) and np.array_equal(f1_g.values, wv)
    want, wuv = cref.gkr_prove(idx, vals, dim, f2, f3, g, threads=cref.max_threads())
    proof = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1, mf2, mf3, g)
    got1 = np.stack([m.evaluations for m in proof.phase1_sumcheck_msgs])
    got2 = np.stack([m.evaluations for m in proof.phase2_sumcheck_msgs])
    assert np.array_equal(got1, want[0]) and np.array_equal(got2, want[1])
    f1_gu = sc.initialize_phase_two(f1_g, wuv[0])
    assert np.array_equal(f1_gu.evaluations, cref.gkr_phase_two(wi, wv, dim, wuv[0]))
    sub = sc.GKRRoundSumcheck.verify(sc.Blake2b512Rng.setup(), dim, proof, proof.extract_sum())
    assert np.array_equal(sub.u, wuv[0]) and np.array_equal(sub.v, wuv[1])
    if dim <= 12:
        assert sub.verify_subclaim(f1, mf2, mf3, g)


def test_gkr_config5_dim20():
    """BASELINE config 5: GKRRoundSumcheck prove, dim=20, 2^20 non-zeros, bit-exact vs the oracle"""
    dim = 20
    idx, vals, f2, f3, g = _random_gkr(dim, 0x5C20241008)
    want, _ = cref.gkr_prove(idx, vals, dim, f2, f3, g, threads=cref.max_threads())
    proof = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), sc.SparseMultilinearExtension(3 * dim, idx, vals),
                                      sc.DenseMultilinearExtension(dim, f2), sc.DenseMultilinearExtension(dim, f3), g)
    assert np.array_equal(np.stack([m.evaluations for m in proof.phase1_sumcheck_msgs]), want[0])
    assert np.array_equal(np.stack([m.evaluations for m in proof.phase2_sumcheck_msgs]), want[1])


def test_gkr_device_resident_inputs_and_interactive_phases():
    """sc_gkr_* with SC_TABLES_ON_DEVICE (inputs read in place from HBM, the initialisation outputs produced in place), and the
    reference's interactive pieces around them: initialize_phase_one -> start_phase1_sumcheck -> dim rounds -> initialize_phase_two
    -> f2.evaluate(u) -> start_phase2_sumcheck -> dim rounds (mod.rs:100-133), every message against the fused proof and the oracle"""
    import torch
    dev = _torch_dev()
    dim = 11
    idx, vals, f2, f3, g = _random_gkr(dim, 777, collide=True)
    td = lambda a: torch.from_numpy(a.view(np.int64)).to(dev)
    f1d = sc.SparseMultilinearExtension(3 * dim, td(idx), td(vals))
    f2d, f3d = sc.DenseMultilinearExtension(dim, td(f2)), sc.DenseMultilinearExtension(dim, td(f3))
    want, wuv = cref.gkr_prove(idx, vals, dim, f2, f3, g, threads=cref.max_threads())
    proof = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1d, f2d, f3d, g)
    assert np.array_equal(np.stack([m.evaluations for m in proof.phase1_sumcheck_msgs]), want[0])
    assert np.array_equal(np.stack([m.evaluations for m in proof.phase2_sumcheck_msgs]), want[1])
    for t, h in ((f1d.values, vals), (f2d.evaluations, f2), (f3d.evaluations, f3)):  # inputs are only read
        assert np.array_equal(t.cpu().numpy().view(np.uint64), h)
    # interactive flow on device-resident tables
    h_g, f1_g = sc.initialize_phase_one(f1d, f3d, g)
    wh, wi, wv = cref.gkr_phase_one(idx, vals, dim, f3, g)
    assert h_g.on_device and f1_g.on_device
    assert np.array_equal(h_g.evaluations.cpu().numpy().view(np.uint64), wh)
    assert np.array_equal(f1_g.indices.cpu().numpy().view(np.uint64), wi) and np.array_equal(f1_g.values.cpu().numpy().view(np.uint64), wv)
    rng = sc.Blake2b512Rng.setup()
    st, u, v_msg = sc.start_phase1_sumcheck(h_g, f2d), [], None
    for i in range(dim):
        m = sc.IPForMLSumcheck.prove_round(st, v_msg)
        assert np.array_equal(m.evaluations, want[0][i]), f"phase 1 round {i + 1}"
        rng.feed(m)
        v_msg = sc.IPForMLSumcheck.sample_round(rng)
        u.append(v_msg.randomness)
    u = np.stack(u)
    assert np.array_equal(u, wuv[0])
    f1_gu = sc.initialize_phase_two(f1_g, u)
    assert np.array_equal(f1_gu.evaluations.cpu().numpy().view(np.uint64), cref.gkr_phase_two(wi, wv, dim, u))
    f2_u = f2d.evaluate(u)
    assert np.array_equal(f2_u, cref.fix_variables(f2, u).reshape(4))
    scaled = sc.gkr_round_sumcheck.scale(f3d, f2_u)
    f2u_int = field.to_int(f2_u)
    assert field.to_ints(scaled.evaluations.cpu().numpy().view(np.uint64)[:50]) == [x * f2u_int % field.P for x in field.to_ints(f3[:50])]
    assert np.array_equal(sc.gkr_round_sumcheck.scale(sc.DenseMultilinearExtension(dim, f3), f2_u).evaluations,
                          scaled.evaluations.cpu().numpy().view(np.uint64))  # host-array flavour of sc_dense_scale
    st2, v_msg = sc.start_phase2_sumcheck(f1_gu, f3d, f2_u), None  # mod.rs:66-82, 122-133
    for i in range(dim):
        m = sc.IPForMLSumcheck.prove_round(st2, v_msg)
        assert np.array_equal(m.evaluations, want[1][i]), f"phase 2 round {i + 1}"
        rng.feed(m)
        v_msg = sc.IPForMLSumcheck.sample_round(rng)
    # a device-resident index out of range is caught by the device-side check
    bad = td(idx).clone()
    bad[3] = 1 << 62
    with pytest.raises(sc.SumcheckError, match="out of range"):
        sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), sc.SparseMultilinearExtension(3 * dim, bad, td(vals)), f2d, f3d, g)


def test_config4_shard_shape_nv25():
    """BASELINE config 4 is nv=28 over 8 GPUs: every GPU holds an nv=25 shard of the 3 tables (3 GiB).  One such shard as a
    stand-alone instance, checked through the size-independent relations (the sharded protocol itself is covered by the gloo
    and logical-shard tests)."""
    nv, shapes, nt = 25, [[0, 1, 2]], 3
    poly, mles, coefs = _device_poly(nv, shapes, nt, 0x5C20241008 + 4)
    proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly, borrow=True)
    sub = sc.MLSumcheck.verify(poly.info(), sc.MLSumcheck.extract_sum(proof), proof)
    assert np.array_equal(state.randomness, sub.point)
    assert np.array_equal(poly.evaluate(sub.point), sub.expected_evaluation)
    # round 1 against the oracle on the low 2^16-entry slices is not meaningful (sums differ); instead pin round 1 by
    # linearity: P(0) + P(1) of the whole instance equals the sum of the two half-instances' claims (high bit 0 / 1)
    import torch
    halves = []
    for h in range(2):
        ph = sc.ListOfProductsOfPolynomials(nv - 1)
        n = 1 << (nv - 1)
        ph.add_product([sc.DenseMultilinearExtension(nv - 1, m.evaluations[h * n:(h + 1) * n]) for m in mles], coefs[0])
        pr = sc.MLSumcheck.prove(ph)
        halves.append(field.to_int(sc.MLSumcheck.extract_sum(pr)))
    assert (halves[0] + halves[1]) % field.P == field.to_int(sc.MLSumcheck.extract_sum(proof))


def test_sharded_rounds_inside_the_library_world1():
    """sc_ml_prove_sharded_rounds with a one-rank RCCL communicator: the all-reduce, publish and fold path of the multi-GPU
    run, compared with the unsharded oracle proof (one rank => no tail)."""
    import torch
    from sumcheck_amd import sharded
    dev = _torch_dev()
    nv, shapes, nt = 18, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
    tabs = [cref.synth_table(66, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(66, 1000, len(shapes))
    d = H.desc_from(nv, shapes, tabs, coefs)
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    engine = sharded.HipShardEngine(nv, shapes, coefs, tabs, dev, borrow=True)
    ncomm = sharded.NativeComm(dev)
    for _ in range(2):  # twice: the handle is rewound, buffers are reused
        engine.reset()
        got, rand = sharded.prove_sharded_native(engine, ncomm, sharded.DistComm(), nv, 4, None)
        assert np.array_equal(got, want) and np.array_equal(rand, wrand)
    ncomm.close()
    # a communicator whose ranks voted against direct publication (policy "rccl_direct" = 0 at sc_comm_init): all-reduce + publish kernel
    with _lib.policy(rccl_direct=0):
        ncomm = sharded.NativeComm(dev)
    before = _lib.plan_stats()["sharded.rccl_publish"]
    engine.reset()
    got, rand = sharded.prove_sharded_native(engine, ncomm, sharded.DistComm(), nv, 4, None)
    assert np.array_equal(got, want) and np.array_equal(rand, wrand) and _lib.plan_stats()["sharded.rccl_publish"] == before + 1
    ncomm.close()


@pytest.mark.parametrize("nv,nt,shapes,device", [
    (1, 2, [[0, 1]], False),
    (2, 3, [[0, 1, 2], [1]], True),
    (3, 2, [[0], [1, 1]], False),
    (4, 3, [[0, 1], [2, 2, 0]], True),
    (5, 3, [[2, 1, 0]], False),
    (7, 4, [[0, 1, 2, 3], [3]], True),
    (12, 3, [[0, 1], [2]], False),
    (13, 40, [[39, 0, 17], [33, 34], [5]], True),   # more tables than one launch takes
    (17, 3, [[0, 1, 2]], True),
])
def test_poly_evaluate_matches_oracle(nv, nt, shapes, device):
    """ListOfProductsOfPolynomials::evaluate (data_structures.rs:99-109) through sc_poly_evaluate: the value and every table's
    evaluation against the oracle's one-variable-at-a-time fix_variables chain (the GPU folds three variables per pass)."""
    tabs = [cref.synth_table(777 + nv, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(777 + nv, 1000, len(shapes))
    point = cref.synth_table(778 + nv, 2000, nv)
    poly, mles = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0" if device else None)
    got, tv = poly.evaluate_with_tables(point)
    d = H.desc_from(nv, shapes, tabs, coefs)
    assert np.array_equal(got, cref.poly_evaluate(d, point))
    for j, m in enumerate(poly.flattened_ml_extensions):  # pool order = order of first use (data_structures.rs:85-93)
        u = next(i for i, x in enumerate(mles) if x is m)
        assert np.array_equal(tv[j], cref.fix_variables(tabs[u], point).reshape(4))


def test_poly_evaluate_rejects_bad_input():
    tabs = [cref.synth_table(5, s, 8) for s in range(2)]
    coefs = cref.synth_table(5, 1000, 1)
    poly, _ = H.hip_poly_from(3, [[0, 1]], tabs, coefs)
    bad = np.tile(np.array([[0xFFFFFFFFFFFFFFFF] * 4], dtype=np.uint64), (3, 1))  # >= p
    with pytest.raises(sc.SumcheckError):
        poly.evaluate(bad)


def _two_rank_worker(rank, world, port, nv, shapes, nt, q):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sumcheck_amd import sharded
        dev = "cuda:0"  # both ranks share the one GPU of the test box; the exchange goes through gloo
        n_loc = (1 << nv) // world
        coefs = cref.synth_table(93, 1000, len(shapes))
        tail = sharded.TailEngines(shapes, coefs, dev)
        out = []
        for seed in (93, 94):  # two proofs: the second one reloads the cached tail prover
            tabs = [cref.synth_table(seed, s, 1 << nv) for s in range(nt)]
            eng = sharded.HipShardEngine(nv - 1, shapes, coefs, [t[rank * n_loc:(rank + 1) * n_loc] for t in tabs], dev, borrow=True)
            out.append(sharded.prove_sharded([eng], sharded.DistComm(), nv, max(len(s) for s in shapes), tail))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_process_sharded_proof_on_one_gpu():
    """One process per shard as under torchrun, world_size 2, with the real HIP engines: both ranks use this box's
    single GPU and exchange through gloo (RCCL refuses two ranks on one device), so everything but the transport is the
    multi-GPU path: per-round widened all-reduce, bind_final + all-gather, cached tail prover, replicated transcript."""
    import socket
    import torch.multiprocessing as mp
    nv, shapes, nt = 13, [[0, 1, 2], [3, 3], [1]], 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, nv, shapes, nt, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    coefs = cref.synth_table(93, 1000, len(shapes))
    for i, seed in enumerate((93, 94)):
        tabs = [cref.synth_table(seed, s, 1 << nv) for s in range(nt)]
        want, wrand = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=cref.max_threads())
        for rank in (0, 1):
            got, rand = res[rank][i]
            assert np.array_equal(got, want)
            assert np.array_equal(rand, wrand)


@pytest.mark.parametrize("num_vars,nnz", [(0, 1), (1, 2), (7, 40), (24, 3000), (60, 5000), (63, 1000)])
def test_sparse_evaluate_matches_big_integers(num_vars, nnz):
    """SparseMultilinearExtension::evaluate through sc_sparse_evaluate against a direct big-integer evaluation of
    sum_i v_i * prod_k (bit_k(i) ? r_k : 1 - r_k)."""
    from sumcheck_amd import field
    rng = np.random.default_rng(1234 + num_vars)
    space = 1 << num_vars
    if space <= 4 * nnz:
        idx = rng.permutation(space)[: min(nnz, space)].astype(np.uint64)
    else:
        idx = np.unique(rng.integers(0, space, size=2 * nnz, dtype=np.uint64))[:nnz]
    vals = cref.synth_table(4321, num_vars, idx.shape[0])
    point = cref.synth_table(4322, num_vars, max(num_vars, 1))[:num_vars]
    f = sc.SparseMultilinearExtension(num_vars, idx, vals)
    got = field.to_int(f.evaluate(point))
    pt = field.to_ints(point) if num_vars else []
    want = 0
    for i, v in zip(idx.tolist(), field.to_ints(vals)):
        w = v
        for k, r in enumerate(pt):
            w = w * (r if (i >> k) & 1 else (1 - r)) % field.P
        want = (want + w) % field.P
    assert got == want


def test_gkr_scratch_cache_release_and_reuse():
    """The GKR entry points keep their scratch and prover handle between calls; sc_release_caches frees them and the next call
    allocates afresh.  Proofs before and after (and across two dimensions, which re-sizes the cache) are identical."""
    def run(dim):
        rng = np.random.default_rng(99 + dim)
        n = 1 << dim
        idx = np.unique(rng.integers(0, 1 << (3 * dim), size=2 * n, dtype=np.uint64))[:n]
        vals, f2, f3, g = cref.synth_table(31, 1, idx.shape[0]), cref.synth_table(31, 2, n), cref.synth_table(31, 3, n), cref.synth_table(31, 4, dim)
        f1 = sc.SparseMultilinearExtension(3 * dim, idx, vals)
        pr = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1, sc.DenseMultilinearExtension(dim, f2), sc.DenseMultilinearExtension(dim, f3), g)
        return np.stack([m.evaluations for m in pr.phase1_sumcheck_msgs + pr.phase2_sumcheck_msgs])
    a10, a7 = run(10), run(7)          # second call shrinks nothing, re-uses the larger arena; the handle is rebuilt for dim 7
    assert sc.lib().sc_release_caches() == 0
    assert np.array_equal(run(10), a10) and np.array_equal(run(7), a7) and np.array_equal(run(7), a7)
    assert sc.lib().sc_release_caches() == 0


@pytest.mark.parametrize("nv,nt,shapes,chunk", [
    (14, 10, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10),   # config-3 shape, 16 chunks
    (15, 5, [[2, 3, 0, 1], [1, 4, 4], [3, 2, 1], [0, 0]], 12),  # shared tables, repeated factors, 8 chunks
    (13, 2, [[0, 1], [1]], 13),                              # one chunk
    (18, 3, [[0, 1, 2]], 0),                                 # default chunk (clamped to the table): one chunk, big rounds after it
    (9, 3, [[0, 1, 2]], 10),                                 # too small to stream: silently the copying handle
    (13, 4, [[0, 0, 1, 2, 3], [1, 2]], 11),                  # five multiplicands: per-product launches per chunk (node-by-node kernel), 4 chunks
    (12, 3, [[0, 1, 2, 0, 1, 2, 0, 1, 2, 0], [2]], 10),      # ten multiplicands: the generic kernel over per-chunk pointer sets
    (12, 4, [[i % 4, (i + 1) % 4] for i in range(14)], 10),   # fourteen products: more than the merged launch takes
])
def test_streamed_host_tables_match_oracle(nv, nt, shapes, chunk):
    """out-of-core mode (sc_prover_init_streamed): the tables stay in host memory and rounds 1 and 2 are computed chunk by chunk
    through a staging ring; every round message, the bound tables after round 2 and whole Fiat-Shamir proofs (twice on the rewound
    handle) against the oracle"""
    tabs = [cref.synth_table(6000 + nv, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(6000 + nv, 1000, len(shapes))
    chal = cref.synth_table(6000 + nv, 2000, nv)
    d = H.desc_from(nv, shapes, tabs, coefs)
    mles = [sc.DenseMultilinearExtension(nv, t) for t in tabs]
    poly = sc.ListOfProductsOfPolynomials(nv)
    for k, sh in enumerate(shapes):
        poly.add_product([mles[i] for i in sh], coefs[k])
    st = sc.IPForMLSumcheck.prover_init(poly, streamed_chunk_log2=chunk)
    op = cref.Prover(d, threads=cref.max_threads())
    v = None
    for i in range(nv):
        want = op.prove_round(None if v is None else v.randomness)
        got = sc.IPForMLSumcheck.prove_round(st, v).evaluations
        assert np.array_equal(got, want), f"round {i + 1}"
        v = sc.VerifierMsg(chal[i])
        if i in (0, 1, 2):
            _, otabs, _ = op.state()
            for u, t in enumerate(st.flattened_ml_extensions):
                assert np.array_equal(t.evaluations, otabs[u]), f"table {u} after round {i + 1}"
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    for _ in range(2):
        st.reset()
        assert np.array_equal(st.prove(sc.Blake2b512Rng.setup()), want)
    assert np.array_equal(st.randomness, wrand)
    for m, t in zip(mles, tabs):  # the streamed inputs are only read
        assert np.array_equal(m.evaluations, t)


def test_cache_limit_and_polling_switch_do_not_change_results():
    """sc_set_cache_limit(0): nothing is kept between calls (every call allocates and frees its own) -- same proofs, same evaluation,
    same GKR proof as with the caches; SC_NO_DEVICE_POLLING / sc_prover_set_polling(p, 0): every round launched after its challenge."""
    nv, nt, shapes = 15, 4, [[0, 1, 2], [3, 3], [1]]
    tabs = [cref.synth_table(7300, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(7300, 1000, len(shapes))
    d = H.desc_from(nv, shapes, tabs, coefs)
    want, wrand = cref.ml_prove(d, threads=4)
    point = cref.synth_table(7300, 2000, nv)
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    try:
        for limit in (0, 1 << 20, 16 << 30):
            _lib.check(sc.lib().sc_set_cache_limit(limit))
            for _ in range(2):
                assert np.array_equal(np.stack([m.evaluations for m in sc.MLSumcheck.prove(poly)]), want)
                assert np.array_equal(poly.evaluate(point), cref.poly_evaluate(d, point))
    finally:
        _lib.check(sc.lib().sc_set_cache_limit(16 << 30))
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    _lib.check(sc.lib().sc_prover_set_polling(st._h, 0))
    assert np.array_equal(st.prove(sc.Blake2b512Rng.setup()), want) and np.array_equal(st.randomness, wrand)
    _lib.check(sc.lib().sc_prover_set_polling(st._h, 1))
    st.reset()
    assert np.array_equal(st.prove(sc.Blake2b512Rng.setup()), want)
    st.close()
    dd, keep = poly._desc(True)
    dd.flags |= _lib.SC_NO_DEVICE_POLLING
    h = C.c_void_p()
    _lib.check(sc.lib().sc_prover_init(C.byref(dd), C.byref(h)))
    proof = np.empty((nv, 4, 4), dtype=np.uint64)
    _lib.check(sc.lib().sc_ml_prove_handle(h, None, C.c_void_p(proof.ctypes.data)))
    sc.lib().sc_prover_free(h)
    assert np.array_equal(proof, want)


@pytest.mark.parametrize("nv,nt,shapes", [(14, 10, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]), (9, 3, [[0, 1, 2], [2, 2]]), (13, 2, [[0, 1]]),
                                          (16, 3, [[0, 1, 2]])])
def test_interactive_rounds_resident_kernel(nv, nt, shapes):
    """IPForMLSumcheck::prove_round round by round (prover.rs:74-77): the late rounds are served by ONE kernel that stays on the GPU
    between calls.  Every message against the oracle's, through every way the dialogue can go: back to back; a verifier that takes
    longer than the kernel's patience (it leaves, the call launches afresh); a state export in the middle (it is asked to leave);
    a misuse error in the middle (MISSING_MSG: the kernel keeps waiting); a reset in the middle; the handle freed while it waits; and
    with the resident kernel switched off (sc_prover_set_resident(p, 0)).  Bound tables and randomness compared where exported."""
    import time
    tabs = [cref.synth_table(7100 + nv, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(7100 + nv, 1000, len(shapes))
    chal = cref.synth_table(7100 + nv, 2000, nv)
    d = H.desc_from(nv, shapes, tabs, coefs)
    op = cref.Prover(d, threads=4)
    want, wtabs = [], {}
    for i in range(nv):
        want.append(op.prove_round(None if i == 0 else chal[i - 1]))
        wtabs[i + 1] = op.state()[1]
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    vmsg = [None] + [sc.VerifierMsg(chal[i]) for i in range(nv - 1)]

    def dialogue(st, pause_at=(), export_at=(), misuse_at=(), stop_after=None):
        got = []
        for i in range(nv):
            if i in pause_at:
                time.sleep(0.02)  # far beyond the patience of ~0.5 ms
            if i in misuse_at and i > 0:
                with pytest.raises(sc.SumcheckError, match="verifier message is empty"):
                    sc.IPForMLSumcheck.prove_round(st, None)
            got.append(sc.IPForMLSumcheck.prove_round(st, vmsg[i]).evaluations)
            assert np.array_equal(got[-1], want[i]), f"round {i + 1}"
            if (i + 1) in export_at:
                for u, t in enumerate(st.flattened_ml_extensions):
                    assert np.array_equal(t.evaluations, wtabs[i + 1][u]), f"table {u} after round {i + 1}"
                assert st.round == i + 1 and np.array_equal(st.randomness, chal[:i])
            if stop_after is not None and i + 1 == stop_after:
                return got
        assert st.round == nv and np.array_equal(st.randomness, chal[: nv - 1])
        for u, t in enumerate(st.flattened_ml_extensions):
            assert np.array_equal(t.evaluations, wtabs[nv][u])
        return got

    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    dialogue(st)                                                     # back to back
    st.reset(); dialogue(st, pause_at={nv - 3, nv - 1})              # the kernel's patience expires twice
    st.reset(); dialogue(st, export_at={max(nv - 5, 1), nv - 2})     # quiesced for a state export, twice
    st.reset(); dialogue(st, export_at={1, 2, 3}, pause_at={2, 5})   # ... and early: before its first bind, while its blocks still hold slices
    st.reset(); dialogue(st, misuse_at={nv - 4, nv - 2})             # errors in between do not disturb it
    st.reset(); dialogue(st, stop_after=nv - 2)                      # abandoned two rounds before the end ...
    st.reset(); dialogue(st)                                         # ... reset while it waits, and proved again
    _lib.check(sc.lib().sc_prover_set_resident(st._h, 0))
    st.reset(); dialogue(st, export_at={nv - 3})                     # without the resident kernel: the launch sequence per round
    _lib.check(sc.lib().sc_prover_set_resident(st._h, 256))
    st.reset(); dialogue(st, stop_after=nv - 1)
    st.close()                                                       # freed while the kernel waits for the last challenge
    st2 = sc.IPForMLSumcheck.prover_init(poly, borrow=True)          # (the pool hands the same handle back)
    dialogue(st2)
    with pytest.raises(sc.SumcheckError, match="Prover is not active"):
        sc.IPForMLSumcheck.prove_round(st2, vmsg[1])
    st2.close()
    with _lib.policy(tail_slices=0):                                 # the resident kernel in its older form: k_tail_rounds, tables through memory
        before = _lib.plan_stats()["resident.rounds"]
        st3 = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
        dialogue(st3)
        st3.reset(); dialogue(st3, pause_at={nv - 2}, export_at={nv - 3})
        st3.close()
        assert _lib.plan_stats()["resident.rounds"] > before


def test_out_of_memory_is_a_status_code_and_a_resident_kernel_does_not_block_other_provers():
    """(1) A handle whose bound-table buffers cannot be allocated (ten tables of 2^34 entries: 4.6 TB) is refused with SC_ERR_OOM, nothing
    crashes, the next call works.  (2) While one handle's resident kernel (interactive protocol) holds the device's tail slot waiting for
    its verifier, another handle proves whole proofs: its late rounds take the pipelined launches instead of waiting for the slot, and
    both get the oracle's bits."""
    import torch
    nv_big, U = 34, 10
    dummy = torch.zeros((16, 4), dtype=torch.int64, device="cuda:0")  # never dereferenced: a borrowing init only allocates
    tabs = (C.c_void_p * U)(*[dummy.data_ptr()] * U)
    coeffs = np.ascontiguousarray(cref.synth_table(1, 1000, 1))
    offs, idx = np.asarray([0, 2], dtype=np.uint32), np.asarray([0, 1], dtype=np.uint32)
    d = _lib.PolyDesc()
    d.num_vars, d.max_multiplicands, d.n_products = nv_big, 2, 1
    d.coeffs = coeffs.ctypes.data_as(C.POINTER(C.c_uint64))
    d.prod_offsets = offs.ctypes.data_as(C.POINTER(C.c_uint32))
    d.prod_indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
    d.n_tables = U
    d.tables = C.cast(tabs, C.POINTER(C.c_void_p))
    d.flags = _lib.SC_TABLES_ON_DEVICE | _lib.SC_TABLES_BORROW
    h = C.c_void_p()
    assert sc.lib().sc_prover_init(C.byref(d), C.byref(h)) == _lib.SC_ERR_OOM and not h.value
    nv, nt, shapes = 13, 4, [[0, 1, 2], [3, 3]]
    tabs = [cref.synth_table(7400, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(7400, 1000, len(shapes))
    dd = H.desc_from(nv, shapes, tabs, coefs)
    want, _ = cref.ml_prove(dd, threads=4)
    chal = cref.synth_table(7400, 2000, nv)
    op = cref.Prover(dd, threads=4)
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    a = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    _lib.check(sc.lib().sc_prover_set_resident(a._h, 1 << 20))  # a patient resident kernel: it holds the slot throughout
    b = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    v = None
    for i in range(nv):
        got = sc.IPForMLSumcheck.prove_round(a, v).evaluations
        assert np.array_equal(got, op.prove_round(None if v is None else v.randomness)), i
        v = sc.VerifierMsg(chal[i])
        if i >= 2:  # a's kernel is resident from its first late round on
            b.reset()
            assert np.array_equal(b.prove(sc.Blake2b512Rng.setup()), want), i
    a.close()
    b.close()


def _stats():
    out = (C.c_uint64 * 8)()
    _lib.check(sc.lib().sc_library_stats(out, 8))
    return dict(zip(("tail_launches", "slot_busy", "slot_reclaims", "resident_starts", "resident_gone", "proof_retries"), [int(x) for x in out]))


def test_tail_slot_of_an_idle_interactive_handle_is_reclaimed_and_pooled_handles_start_from_default_policy():
    """ADVICE r3.  (1) An interactive handle that sits idle mid-protocol keeps the device's tail slot only while its resident kernel is
    actually on the GPU: once the kernel's patience (~0.5 ms) has expired another prover TAKES THE SLOT OVER (sc_library_stats: a
    reclaim, a persistent-tail launch, no 'slot busy') instead of falling back to pipelined launches for as long as the idle handle
    lives; the idle handle then finds its kernel gone and carries on through the ordinary path with the right messages.
    (2) What one owner set on a handle (the resident kernel's patience) does not reach the next owner through the handle pool."""
    import time
    nv, nt, shapes = 12, 4, [[0, 1, 2], [3, 3]]
    tabs = [cref.synth_table(7500, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(7500, 1000, len(shapes))
    dd = H.desc_from(nv, shapes, tabs, coefs)
    want, _ = cref.ml_prove(dd, threads=4)
    chal = cref.synth_table(7500, 2000, nv)
    op = cref.Prover(dd, threads=4)
    wmsgs = [op.prove_round(None if i == 0 else chal[i - 1]) for i in range(nv)]  # (first: the dialogue below must not pause for the CPU)
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    a = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    b = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    v = None
    for i in range(nv):
        got = sc.IPForMLSumcheck.prove_round(a, v).evaluations
        assert np.array_equal(got, wmsgs[i]), i
        v = sc.VerifierMsg(chal[i])
        if i == 3:  # a's resident kernel is on the GPU (it started with the first late round); let its patience run out
            assert _stats()["resident_starts"] >= 1
            time.sleep(0.05)
            s0 = _stats()
            b.reset()
            assert np.array_equal(b.prove(sc.Blake2b512Rng.setup()), want)
            s1 = _stats()
            assert s1["slot_reclaims"] == s0["slot_reclaims"] + 1 and s1["tail_launches"] == s0["tail_launches"] + 1 and s1["slot_busy"] == s0["slot_busy"], (s0, s1)
    assert _stats()["resident_gone"] >= 1  # a noticed on its next call and took the ordinary path: every message above was still right
    a.close()
    b.close()
    # (2) the pool: sc_ml_prove builds (or takes) a handle of this shape, an owner makes its resident kernel very patient, frees it;
    # the next owner of the pooled handle gets the default patience back: its idle kernel is gone after ~0.5 ms, not after seconds
    dsc, keep = poly._desc(True)
    h = C.c_void_p()
    _lib.check(sc.lib().sc_prover_init(C.byref(dsc), C.byref(h)))
    _lib.check(sc.lib().sc_prover_set_resident(h, 1 << 20))
    sc.lib().sc_prover_free(h)  # offered to the pool
    h2 = C.c_void_p()
    _lib.check(sc.lib().sc_prover_init(C.byref(dsc), C.byref(h2)))
    out = np.empty((4, 4), dtype=np.uint64)
    ch = [np.ascontiguousarray(chal[i]) for i in range(nv)]  # (kept alive: the calls below take raw pointers)
    for i in range(4):
        _lib.check(sc.lib().sc_prove_round(h2, C.c_void_p(ch[i - 1].ctypes.data) if i else None, C.c_void_p(out.ctypes.data)))
    g0 = _stats()["resident_gone"]
    time.sleep(0.05)  # default patience: the kernel has left by now (with 2^20 polls it would still be there)
    _lib.check(sc.lib().sc_prove_round(h2, C.c_void_p(ch[3].ctypes.data), C.c_void_p(out.ctypes.data)))
    assert _stats()["resident_gone"] == g0 + 1
    sc.lib().sc_prover_free(h2)


def test_provers_on_several_threads_share_one_gpu():
    """Three host threads, one GPU: two whole-proof provers (pipelined late rounds + persistent tail kernel) and a GKR prover,
    each repeating its proof.  The library serialises its HIP calls per device (prover_internal.hpp: DeviceGate) so that a kernel waiting
    for one thread's challenge never sits in front of another thread's blocked call; every proof must equal the oracle's."""
    import threading

    def ml_worker(k, nv, shapes, nt, reps, out):
        try:
            tabs = [cref.synth_table(3100 + k, s, 1 << nv) for s in range(nt)]
            coefs = cref.synth_table(3100 + k, 1000, len(shapes))
            want, _ = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=4)
            poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
            st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
            bad = 0
            for _ in range(reps):
                st.reset()
                bad += not np.array_equal(np.asarray(st.prove()).reshape(want.shape), want)
            out[k] = bad
        except Exception as e:  # a give-up ("proof is void") lands here
            out[k] = repr(e)

    def gkr_worker(k, dim, reps, out):
        try:
            n = 1 << dim
            rng = np.random.default_rng(77)
            idx = np.unique(rng.integers(0, 1 << (3 * dim), size=2 * n, dtype=np.uint64))
            vals, f2, f3, g = (cref.synth_table(3200, 1, idx.shape[0]), cref.synth_table(3200, 2, n), cref.synth_table(3200, 3, n),
                               cref.synth_table(3200, 4, dim))
            want, _ = cref.gkr_prove(idx, vals, dim, f2, f3, g, threads=4)
            f1 = sc.SparseMultilinearExtension(3 * dim, idx, vals)
            bad = 0
            for _ in range(reps):
                pr = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1, sc.DenseMultilinearExtension(dim, f2),
                                               sc.DenseMultilinearExtension(dim, f3), g)
                bad += not (np.array_equal(np.stack([m.evaluations for m in pr.phase1_sumcheck_msgs]), want[0])
                            and np.array_equal(np.stack([m.evaluations for m in pr.phase2_sumcheck_msgs]), want[1]))
            out[k] = bad
        except Exception as e:
            out[k] = repr(e)

    out = [None, None, None]
    ts = [threading.Thread(target=ml_worker, args=(0, 15, [[0, 1, 2], [3]], 4, 200, out)),
          threading.Thread(target=ml_worker, args=(1, 19, [[0, 1, 2, 3], [1, 2]], 4, 60, out)),
          threading.Thread(target=gkr_worker, args=(2, 12, 40, out))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert out == [0, 0, 0], out


@pytest.mark.parametrize("case", ["sorted", "shuffled", "crowded_x", "crowded_y", "duplicates", "tiny_dim", "list_form"])
def test_gkr_prove_bucketed_initialisation(case, request):
    """sc_gkr_prove builds a_hg and f1(g,u,.) by bucketing the non-zeros and adding terms in LDS (gkr.hip: k_bucket_accumulate)
    instead of sorting and merging.  Same proof bits as the oracle for: index-ordered input (phase two skips its radix pass),
    shuffled input, index distributions that crowd one x or y bucket (fallback to the list form, in either phase), repeated
    indices (summed, as the list form does), dim below the bucket width, and the list form forced by sc_set_policy("gkr_direct", 0)."""
    dim = 4 if case == "tiny_dim" else 14
    n = 1 << dim
    rng = np.random.default_rng(99)
    z = rng.integers(0, n, size=4 * n, dtype=np.uint64)
    x = rng.integers(0, n, size=4 * n, dtype=np.uint64)
    y = rng.integers(0, n, size=4 * n, dtype=np.uint64)
    if case == "crowded_x":
        x &= np.uint64(7)  # every non-zero in x bucket 0 (16 cells per bucket at dim 14)
    if case == "crowded_y":
        y &= np.uint64(7)
    idx = z | (x << np.uint64(dim)) | (y << np.uint64(2 * dim))
    if case != "duplicates":
        idx = np.unique(idx)
    else:
        idx = np.concatenate([np.unique(idx), idx[:1000]])
    if case in ("shuffled", "crowded_x", "duplicates"):
        idx = idx[rng.permutation(idx.shape[0])]
    vals, f2, f3, g = (cref.synth_table(77, 1, idx.shape[0]), cref.synth_table(77, 2, n), cref.synth_table(77, 3, n), cref.synth_table(77, 4, dim))
    if case == "duplicates":  # the oracle takes a map: give it the merged list
        order = np.argsort(idx, kind="stable")
        si, sv = idx[order], vals[order]
        ui, start = np.unique(si, return_index=True)
        ints = field.to_ints(sv)
        merged = []
        for a, b in zip(start, list(start[1:]) + [len(si)]):
            merged.append(sum(ints[a:b]) % field.P)
        oi, ov = ui, H.mont(merged)
    else:
        oi, ov = idx, vals
    if case == "list_form":  # sort + merge instead of the bucketed kernels
        request.addfinalizer(lambda: _lib.set_policy("gkr_direct", 1))
        _lib.set_policy("gkr_direct", 0)
    want, wuv = cref.gkr_prove(oi, ov, dim, f2, f3, g, threads=cref.max_threads())
    for on_device in (False, True):
        if on_device:
            import torch
            dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()
            f1 = sc.SparseMultilinearExtension(3 * dim, dev(idx), dev(vals))
            m2, m3 = sc.DenseMultilinearExtension(dim, dev(f2)), sc.DenseMultilinearExtension(dim, dev(f3))
        else:
            f1 = sc.SparseMultilinearExtension(3 * dim, idx, vals)
            m2, m3 = sc.DenseMultilinearExtension(dim, f2), sc.DenseMultilinearExtension(dim, f3)
        proof = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1, m2, m3, g)
        assert np.array_equal(np.stack([m.evaluations for m in proof.phase1_sumcheck_msgs]), want[0]), (case, on_device)
        assert np.array_equal(np.stack([m.evaluations for m in proof.phase2_sumcheck_msgs]), want[1]), (case, on_device)
        # the stand-alone initialisations take the same bucketed route for their dense tables (and skip the sort of an ordered list)
        wh, wi, wv = cref.gkr_phase_one(oi, ov, dim, f3, g)
        h_g, f1_g = sc.initialize_phase_one(f1, m3, g)
        host = lambda a: a.cpu().numpy().view(np.uint64) if on_device else a
        assert np.array_equal(host(h_g.evaluations), wh) and np.array_equal(host(f1_g.indices).reshape(-1), wi) and np.array_equal(host(f1_g.values), wv), (case, on_device)
        back = f1_g if case == "sorted" else sc.SparseMultilinearExtension(2 * dim, f1_g.indices[::-1].copy() if not on_device else f1_g.indices.flip(0).contiguous(),
                                                                             f1_g.values[::-1].copy() if not on_device else f1_g.values.flip(0).contiguous())
        assert np.array_equal(host(sc.initialize_phase_two(back, wuv[0]).evaluations), cref.gkr_phase_two(wi, wv, dim, wuv[0])), (case, on_device)


def test_one_shot_proofs_reuse_the_kept_prover():
    """MLSumcheck.prove in a loop (sc_ml_prove with no state returned): the library keeps the prover it built and rewinds it onto the
    next polynomial of the same structure -- different tables each time, host and device tables, a different structure in between,
    a state handed out and freed -- every proof against the oracle; sc_release_caches in the middle."""
    import torch
    nv = 12
    shapes_a, shapes_b = [[0, 1, 2], [1, 3]], [[0, 1], [2, 2, 3]]
    for it, (shapes, dev) in enumerate([(shapes_a, "cuda:0"), (shapes_a, "cuda:0"), (shapes_a, None), (shapes_b, "cuda:0"), (shapes_a, "cuda:0"),
                                        (shapes_a, None), (shapes_a, None)]):
        tabs = [cref.synth_table(600 + it, s, 1 << nv) for s in range(4)]
        coefs = cref.synth_table(600, 1000, len(shapes))  # same coefficients: same structure
        want, wrand = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=4)
        poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device=dev)
        got = np.stack([m.evaluations for m in sc.MLSumcheck.prove(poly)])
        assert np.array_equal(got, want), it
        if it == 1:  # a state that leaves and comes back through sc_prover_free
            proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly)
            assert np.array_equal(np.stack([m.evaluations for m in proof]), want) and np.array_equal(state.randomness, wrand)
            del state
        if it == 4:
            _lib.check(sc.lib().sc_release_caches())
    torch.cuda.synchronize()


TAIL_SHAPES = [
    (12, 10, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]),  # 2048 pairs: 64 blocks, six hand-over rounds, then block 0 alone
    (13, 2, [[0, 1]]),                                 # a GKR phase's shape; the tail starts behind a pipelined round (its first round binds)
    (8, 3, [[0, 1, 2]]),                               # 128 pairs: the block count follows the tail's length
    (5, 12, [[0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10], [11, 11]]),  # one block from the start; eight multiplicands, a repeated table
    (16, 7, [[0, 1, 2, 3], [4, 5, 6], [1, 1], [2], [3, 3, 3], [5, 6, 6, 0]]),  # tables arrive from the big rounds in the internal format
]


@pytest.mark.parametrize("nv,nt,shapes", TAIL_SHAPES)
def test_tail_with_tables_resident_in_lds(nv, nt, shapes):
    """k_tail_slices (kernels_tail.hip): the latency-bound rounds of a whole proof out of LDS.  Whole Fiat-Shamir proofs against the
    oracle, the state the tail leaves behind (randomness, the final two-entry tables: written back from LDS), the library's count of
    such launches -- and the same proof through k_tail_rounds (policy "tail_slices" = 0) to the same bits."""
    tabs = [cref.synth_table(4242 + nv, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(4242 + nv, 1000, len(shapes))
    d = H.desc_from(nv, shapes, tabs, coefs)
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    stats = (C.c_uint64 * 8)()
    _lib.check(sc.lib().sc_library_stats(stats, 8))
    before = (stats[0], stats[6])
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    for rep in range(3):  # (the handle's tags only grow: a second and third proof on the rewound handle)
        st.reset()
        proof = st.prove()
        assert np.array_equal(np.asarray(proof).reshape(want.shape), want), f"proof {rep}"
    _lib.check(sc.lib().sc_library_stats(stats, 8))
    assert stats[0] - before[0] == 3 and stats[6] - before[1] == 3, (list(stats), before)  # every tail ran out of LDS
    op = cref.Prover(d, threads=cref.max_threads())
    v = None
    for i in range(nv):
        op.prove_round(v)
        v = wrand[i]
    _, otabs, _ = op.state()
    assert np.array_equal(st.randomness, wrand)
    for u, t in enumerate(st.flattened_ml_extensions):
        assert np.array_equal(t.evaluations, otabs[u]), f"final table {u}"
    # ... and through k_tail_rounds (policy "tail_slices" = 0: a fresh handle, the same tables)
    with _lib.policy(tail_slices=0):
        _lib.check(sc.lib().sc_library_stats(stats, 8))
        before = (stats[0], stats[6])
        plans = _lib.plan_stats()
        proof = sc.MLSumcheck.prove(poly)
        _lib.check(sc.lib().sc_library_stats(stats, 8))
        assert stats[0] - before[0] == 1 and stats[6] == before[1], (list(stats), before)
        assert _lib.plan_stats()["tail.rounds"] == plans["tail.rounds"] + 1
    assert np.array_equal(np.stack([m.evaluations for m in proof]), want)


WIDE_SHAPES = [
    (17, 5, [[0, 1, 2, 3, 4]]),
    (17, 6, [[0, 1, 2, 3, 4, 5]]),
    (17, 7, [[0, 1, 2, 3, 4, 5, 6]]),
    (17, 8, [[0, 1, 2, 3, 4, 5, 6, 7]]),
    (18, 13, [[0, 1, 2, 3, 4], [5, 6, 7, 8, 9, 10, 11, 12], [2, 2, 3, 9, 9, 1], [4, 4, 4, 4, 4, 4, 4], [12]]),  # repeated factors, shared tables, a mix of lengths
]


WIDE_SHAPES += [
    (16, 9, [[0, 1, 2, 3, 4, 5, 6, 7, 8]]),
    (16, 10, [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9]]),
    (16, 11, [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]]),
    (16, 12, [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]]),
    (17, 7, [[0, 1, 2, 3, 4, 5, 6, 0, 1, 2, 3, 4], [6, 6, 6, 6, 6, 6, 6, 6, 6, 5], [1, 2, 3], [0, 1, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 1]]),  # 12 / 10 with repeats, a short one, and 13 (node by node)
    # more tables than one launch's arguments hold (40 > kMaxSmallTables): test_normal_polynomial's shape, every product over its own tables
    (16, 40, [list(range(0, 4)), list(range(4, 10)), list(range(10, 18)), list(range(18, 28)), list(range(28, 40))]),
    (15, 40, [list(range(8 * k, 8 * k + 8)) for k in range(5)]),
    # nine-to-twelve trees that share tables with each other and with shorter products, with repeats, behind one bind pass per round
    (17, 12, [list(range(12)), list(range(9)), [3] * 10, [1, 2, 3, 4], [5, 6, 7, 8, 9, 10, 11, 11], [11, 10, 0, 0, 9, 9, 9, 4, 2, 2, 7]]),
]


@pytest.mark.parametrize("nv,nt,shapes", WIDE_SHAPES)
def test_wide_products_product_tree_with_node_extension(nv, nt, shapes):
    """k_prod_tree_wide<5..8> (kernels_wide.hip) and k_prod_tree_wide16<9..12> (kernels_wide16.hip): the big rounds of products of five
    to twelve multiplicands -- the reference's own test shapes (ml_sumcheck/test.rs:122-167: 4..12 per product) at a size that runs
    them -- round by round against the oracle with fixed challenges (every message and the bound tables), then whole Fiat-Shamir proofs."""
    tabs = [cref.synth_table(9090 + nv, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(9090 + nv, 1000, len(shapes))
    chal = cref.synth_table(9090 + nv, 2000, nv)
    d = H.desc_from(nv, shapes, tabs, coefs)
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
    op = cref.Prover(d, threads=cref.max_threads())
    st = sc.IPForMLSumcheck.prover_init(poly)
    v = None
    for i in range(nv):
        want = op.prove_round(None if v is None else v.randomness)
        got = sc.IPForMLSumcheck.prove_round(st, v).evaluations
        assert np.array_equal(got, want), f"round {i + 1}"
        v = sc.VerifierMsg(chal[i])
    _, otabs, _ = op.state()
    for u, t in enumerate(st.flattened_ml_extensions):
        assert np.array_equal(t.evaluations, otabs[u])
    proof = sc.MLSumcheck.prove(poly)
    want, _ = cref.ml_prove(d, threads=cref.max_threads())
    assert np.array_equal(np.stack([m.evaluations for m in proof]), want)
    if (nv, nt) in ((17, 6), (16, 12)):  # ... and node by node (policy "wide_tree" = 0: k_prod_round_fe up to eight, k_sum_generic beyond), a handle built under it
        with _lib.policy(wide_tree=0):
            before = _lib.plan_stats()
            proof = sc.MLSumcheck.prove(poly)
            after = _lib.plan_stats()
        assert np.array_equal(np.stack([m.evaluations for m in proof]), want)
        key = "big.node_by_node" if nt <= 8 else "big.generic"
        assert after[key] > before[key] and after["big.wide"] == before["big.wide"] and after["big.wide16"] == before["big.wide16"]


@pytest.mark.parametrize("nv", [22, 19])
def test_staged_init_host_tables_round_one_under_the_copy(nv):
    """sc_prover_init over HOST tables of a merged-kernel shape (protocol.hip: staged_copy_and_round1): the tables go in in chunks and round 1
    is computed under the copy -- IPForMLSumcheck::prover_init's deep copy (prover.rs:55-59) and the first prove_round in one pass.  At
    nv = 22 (config 3's shape; chunks of 1/2 ... 1/32, 1/32 on two copy streams) and nv = 19 (1/2, 1/4, 1/4): the interactive rounds with bound tables, whole Fiat-Shamir proofs (one-shot and on a handle),
    a reset onto OTHER tables (the pool's path), the caller's arrays untouched and droppable after init -- all against the oracle, and the
    same with the staged form switched off."""
    shapes, nt = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
    tabs = [cref.synth_table(6100, s, 1 << nv) for s in range(nt)]
    tabs2 = [cref.synth_table(6200, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(6100, 1000, len(shapes))
    chal = cref.synth_table(6100, 2000, nv)
    d = H.desc_from(nv, shapes, tabs, coefs)
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    want2, _ = cref.ml_prove(H.desc_from(nv, shapes, tabs2, coefs), threads=cref.max_threads())
    op = cref.Prover(d, threads=cref.max_threads())
    for staged in (1, 0):
        with _lib.policy(staged_init=staged):
            before = _lib.plan_stats()["big.staged_round1"]
            copies = [t.copy() for t in tabs]
            poly, _ = H.hip_poly_from(nv, shapes, copies, coefs)
            st = sc.IPForMLSumcheck.prover_init(poly)  # host tables, copying handle
            for t in copies:
                t[:] = 0  # prover_init has copied: the caller's memory is its own again
            assert _lib.plan_stats()["big.staged_round1"] == before + staged
            if staged:  # the interactive rounds, message by message, and the bound tables after round 3
                v = None
                for i in range(4):
                    w = op.prove_round(None if v is None else v.randomness)
                    got = sc.IPForMLSumcheck.prove_round(st, v).evaluations
                    assert np.array_equal(got, w), f"round {i + 1}"
                    v = sc.VerifierMsg(chal[i])
                _, otabs, _ = op.state()
                for u, t in enumerate(st.flattened_ml_extensions):
                    assert np.array_equal(t.evaluations, otabs[u]), f"table {u}"
            st.close()
            poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs)
            proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly)  # a handle out of the pool: reset onto these tables
            assert np.array_equal(np.stack([m.evaluations for m in proof]), want) and np.array_equal(state.randomness, wrand)
            del state
            poly2, _ = H.hip_poly_from(nv, shapes, tabs2, coefs)
            assert np.array_equal(np.stack([m.evaluations for m in sc.MLSumcheck.prove(poly2)]), want2)
            assert _lib.plan_stats()["big.staged_round1"] >= before + 3 * staged
