"""pyoracle -- big-integer CPU restatement of the arkworks-rs/sumcheck prover path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product (`sumcheck_amd/`, `bench.py`'s
timed legs) may import this file; only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may, and only as the checker.

This is the *independent* second oracle (the first is `oracle/oracle.c`).  It uses
Python integers reduced `% P` and `hashlib.blake2b`, i.e. it shares no arithmetic
or hashing code with the HIP library or with the C oracle.  Elements are kept as
canonical integers in [0, P); the 4 x u64 Montgomery limb form that arkworks keeps
in memory (`ark_ff::Fp<MontBackend<FrConfig,4>,4>`) appears only at the boundary
helpers `to_mont_limbs` / `from_mont_limbs`.

PARITY STATUS ("parity unpinned" for one layer):
  * The reference is Rust and neither cargo nor rustc exists in the build image, and
    its tests hold no golden vectors.  Field arithmetic is
    pinned by published BLS12-381 constants (R, R^2, INV recomputed below and by
    Fermat checks), the hash by the RFC 7693 "abc" known answer, the protocol by
    the algebraic relations the reference's own tests assert (extract_sum == true
    sum, verifier acceptance, final evaluate(point) check) and by the one literal
    known answer in the tree (interpolate_uni_poly([0,1,4,9], 3) == 9,
    reference src/ml_sumcheck/protocol/verifier.rs:327-331).
  * `F::rand`, `CanonicalSerialize` and `SparseMultilinearExtension::fix_variables`
    live in ark-ff / ark-serialize / ark-poly (algebra master @ 2024-10, declared
    0.4.0, reference Cargo.toml:19-22,61-67), whose source is not in /root/reference.
    They are restated from their published behaviour; that layer is UNPINNED against
    a cargo build.

Every function cites the reference file:line it follows (paths relative to the
reference repository root).
"""
from __future__ import annotations

import hashlib
import struct
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

# ---------------------------------------------------------------------------------------------
# BLS12-381 scalar field (ark-test-curves bls12_381::Fr; reference src/ml_sumcheck/test.rs:13)
# ---------------------------------------------------------------------------------------------
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % P
R2 = (R * R) % P
RINV = pow(R, -1, P)
MASK64 = (1 << 64) - 1


def to_mont_limbs(x: int) -> Tuple[int, int, int, int]:
    """canonical int -> 4 x u64 little-endian limbs of x*R mod P (ark-ff MontBackend layout)."""
    m = (x * R) % P
    return tuple((m >> (64 * i)) & MASK64 for i in range(4))  # type: ignore[return-value]


def from_mont_limbs(l: Sequence[int]) -> int:
    """4 x u64 Montgomery limbs -> canonical int."""
    m = sum(int(l[i]) << (64 * i) for i in range(4))
    assert m < P, "non-canonical Montgomery representation"
    return (m * RINV) % P


# ---------------------------------------------------------------------------------------------
# ark-poly DenseMultilinearExtension (external; Appendix B of SURVEY.md)
# ---------------------------------------------------------------------------------------------
def dense_fix_variables(evals: Sequence[int], point: Sequence[int]) -> List[int]:
    """DenseMultilinearExtension::fix_variables(partial_point): binds variables LSB first.

    poly[b] = poly[2b] + r*(poly[2b+1] - poly[2b]).  Used at reference
    src/ml_sumcheck/protocol/prover.rs:88 with a single challenge.
    """
    poly = list(evals)
    for r in point:
        half = len(poly) // 2
        poly = [(poly[2 * b] + r * (poly[2 * b + 1] - poly[2 * b])) % P for b in range(half)]
    return poly


def dense_evaluate(evals: Sequence[int], point: Sequence[int]) -> int:
    """DenseMultilinearExtension::evaluate(point) == fix_variables(point)[0]."""
    assert len(evals) == 1 << len(point)
    return dense_fix_variables(evals, point)[0]


# ---------------------------------------------------------------------------------------------
# ListOfProductsOfPolynomials (reference src/ml_sumcheck/data_structures.rs:25-110)
# ---------------------------------------------------------------------------------------------
class ListOfProductsOfPolynomials:
    """Tables are Python lists; de-duplication is by object identity (`id`), mirroring the
    reference's `Rc::as_ptr` lookup (data_structures.rs:85-93)."""

    def __init__(self, num_variables: int):  # data_structures.rs:59-67
        self.max_multiplicands = 0
        self.num_variables = num_variables
        self.products: List[Tuple[int, List[int]]] = []
        self.flattened_ml_extensions: List[List[int]] = []
        self._lookup: Dict[int, int] = {}

    def add_product(self, product: Iterable[List[int]], coefficient: int) -> None:
        # data_structures.rs:71-96
        product = list(product)
        assert len(product) > 0
        self.max_multiplicands = max(self.max_multiplicands, len(product))
        indexed = []
        for m in product:
            assert len(m) == 1 << self.num_variables, (
                "product has a multiplicand with wrong number of variables")
            key = id(m)
            if key in self._lookup:
                indexed.append(self._lookup[key])
            else:
                idx = len(self.flattened_ml_extensions)
                self.flattened_ml_extensions.append(m)
                self._lookup[key] = idx
                indexed.append(idx)
        self.products.append((coefficient % P, indexed))

    def info(self) -> Tuple[int, int]:  # data_structures.rs:39-44 -> (max_multiplicands, num_variables)
        return (self.max_multiplicands, self.num_variables)

    def evaluate(self, point: Sequence[int]) -> int:  # data_structures.rs:99-109
        acc = 0
        for c, idxs in self.products:
            prod = c
            for i in idxs:
                prod = prod * dense_evaluate(self.flattened_ml_extensions[i], point) % P
            acc = (acc + prod) % P
        return acc


# ---------------------------------------------------------------------------------------------
# Prover (reference src/ml_sumcheck/protocol/prover.rs)
# ---------------------------------------------------------------------------------------------
class ProverState:  # prover.rs:19-33
    def __init__(self):
        self.randomness: List[int] = []
        self.list_of_products: List[Tuple[int, List[int]]] = []
        self.flattened_ml_extensions: List[List[int]] = []
        self.num_vars = 0
        self.max_multiplicands = 0
        self.round = 0


def prover_init(poly: ListOfProductsOfPolynomials) -> ProverState:
    """prover.rs:49-69"""
    if poly.num_variables == 0:
        raise RuntimeError("Attempt to prove a constant.")
    st = ProverState()
    st.flattened_ml_extensions = [list(t) for t in poly.flattened_ml_extensions]  # deep copy, 55-59
    st.list_of_products = [(c, list(ix)) for c, ix in poly.products]
    st.num_vars = poly.num_variables
    st.max_multiplicands = poly.max_multiplicands
    st.round = 0
    return st


def prove_round(st: ProverState, v_msg: Optional[int]) -> List[int]:
    """prover.rs:74-153 (literal loop nest, including the start += step sequence)."""
    if v_msg is not None:
        if st.round == 0:
            raise RuntimeError("first round should be prover first.")
        st.randomness.append(v_msg % P)
        r = st.randomness[st.round - 1]
        st.flattened_ml_extensions = [dense_fix_variables(t, [r]) for t in st.flattened_ml_extensions]
    elif st.round > 0:
        raise RuntimeError("verifier message is empty")
    st.round += 1
    if st.round > st.num_vars:
        raise RuntimeError("Prover is not active")
    i, nv, degree = st.round, st.num_vars, st.max_multiplicands
    products_sum = [0] * (degree + 1)
    for b in range(1 << (nv - i)):
        for coefficient, products in st.list_of_products:
            product = [coefficient] * (degree + 1)
            for j in products:
                table = st.flattened_ml_extensions[j]
                start = table[b << 1]
                step = (table[(b << 1) + 1] - start) % P
                for t in range(degree + 1):
                    product[t] = product[t] * start % P
                    start = (start + step) % P
            for t in range(degree + 1):
                products_sum[t] = (products_sum[t] + product[t]) % P
    return products_sum


# ---------------------------------------------------------------------------------------------
# Verifier (reference src/ml_sumcheck/protocol/verifier.rs) -- the parity checker
# ---------------------------------------------------------------------------------------------
def interpolate_uni_poly(p_i: Sequence[int], eval_at: int) -> int:
    """verifier.rs:139-251.  The three machine-integer tiers of the reference compute the same
    field value; this restatement uses the plain Lagrange form (field arithmetic is exact)."""
    n = len(p_i)
    x = eval_at % P
    if x < n:  # verifier.rs:152-164 early return when eval_at is one of the nodes
        return p_i[x] % P
    res = 0
    for i in range(n):
        num, den = 1, 1
        for j in range(n):
            if j != i:
                num = num * (x - j) % P
                den = den * (i - j) % P
        res = (res + p_i[i] * num % P * pow(den, -1, P)) % P
    return res


class Reject(Exception):
    pass


def check_and_generate_subclaim(nv: int, max_multiplicands: int, polys: Sequence[Sequence[int]],
                                randomness: Sequence[int], asserted_sum: int) -> Tuple[List[int], int]:
    """verifier.rs:90-121"""
    if len(polys) != nv:
        raise RuntimeError("insufficient rounds")
    expected = asserted_sum % P
    for i in range(nv):
        ev = polys[i]
        if len(ev) != max_multiplicands + 1:
            raise RuntimeError("incorrect number of evaluations")
        if (ev[0] + ev[1]) % P != expected:
            raise Reject("Prover message is not consistent with the claim.")
        expected = interpolate_uni_poly(ev, randomness[i])
    return list(randomness), expected


# ---------------------------------------------------------------------------------------------
# Transcript (reference src/rng.rs) + ark-serialize / ark-ff sampling semantics (UNPINNED layer)
# ---------------------------------------------------------------------------------------------
class Blake2b512Rng:
    def __init__(self):  # rng.rs:30-34
        self.d = hashlib.blake2b(digest_size=64)

    def feed_bytes(self, buf: bytes) -> None:  # rng.rs:36-41 after serialize_uncompressed
        self.d.update(buf)

    def fill_bytes(self, n: int) -> bytes:  # rng.rs:61-80
        out = self.d.copy().digest()
        dest = bytearray()
        dptr = 0
        while len(dest) < n:
            dest.append(out[dptr])
            dptr += 1
            if dptr == 64:
                self.d.update(out)
                out = self.d.copy().digest()
                dptr = 0
        self.d.update(out)  # rng.rs:78 -- always, even if `out` was only partly consumed
        return bytes(dest)

    def next_u64(self) -> int:  # rng.rs:51-55
        return struct.unpack("<Q", self.fill_bytes(8))[0]


def ser_fr(x: int) -> bytes:
    """ark-serialize: Fp -> 32 bytes LE of the canonical (non-Montgomery) integer."""
    return int(x % P).to_bytes(32, "little")


def ser_prover_msg(evals: Sequence[int]) -> bytes:
    """ProverMsg{evaluations: Vec<F>} (prover.rs:13-17): u64 LE length then elements."""
    return struct.pack("<Q", len(evals)) + b"".join(ser_fr(e) for e in evals)


def ser_poly_info(max_multiplicands: int, num_variables: int) -> bytes:
    """PolynomialInfo (data_structures.rs:47-55): two usize as u64 LE, in field order."""
    return struct.pack("<QQ", max_multiplicands, num_variables)


def sample_fr(rng: Blake2b512Rng) -> int:
    """sample_round (verifier.rs:128-131) = F::rand(rng).  ark-ff Fp sampling: draw 4 x next_u64
    as LE limbs, clear the top (256-255)=1 bit, accept iff < P; the limbs ARE the internal
    Montgomery representation, so the sampled element is limbs * R^-1."""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= MASK64 >> 1
        m = sum(limbs[i] << (64 * i) for i in range(4))
        if m < P:
            return (m * RINV) % P


# ---------------------------------------------------------------------------------------------
# MLSumcheck drivers (reference src/ml_sumcheck/mod.rs)
# ---------------------------------------------------------------------------------------------
def extract_sum(proof: Sequence[Sequence[int]]) -> int:  # mod.rs:26-28
    return (proof[0][0] + proof[0][1]) % P


def ml_prove_as_subprotocol(rng: Blake2b512Rng, poly: ListOfProductsOfPolynomials):
    """mod.rs:50-70"""
    rng.feed_bytes(ser_poly_info(*poly.info()))
    st = prover_init(poly)
    v = None
    msgs = []
    for _ in range(poly.num_variables):
        pm = prove_round(st, v)
        rng.feed_bytes(ser_prover_msg(pm))
        msgs.append(pm)
        v = sample_fr(rng)
    st.randomness.append(v)  # mod.rs:65-67: recorded, never bound
    return msgs, st


def ml_prove(poly: ListOfProductsOfPolynomials):  # mod.rs:42-45
    return ml_prove_as_subprotocol(Blake2b512Rng(), poly)[0]


def ml_verify_as_subprotocol(rng: Blake2b512Rng, info: Tuple[int, int], claimed_sum: int,
                             proof: Sequence[Sequence[int]]):
    """mod.rs:84-100 (verify_round = store + sample, verifier.rs:54-83)."""
    max_mult, nv = info
    rng.feed_bytes(ser_poly_info(max_mult, nv))
    rs = []
    for i in range(nv):
        rng.feed_bytes(ser_prover_msg(proof[i]))
        rs.append(sample_fr(rng))
    return check_and_generate_subclaim(nv, max_mult, proof, rs, claimed_sum)


def ml_verify(info, claimed_sum, proof):  # mod.rs:73-80
    return ml_verify_as_subprotocol(Blake2b512Rng(), info, claimed_sum, proof)


# ---------------------------------------------------------------------------------------------
# ark-poly SparseMultilinearExtension (external; Appendix B of SURVEY.md)
# ---------------------------------------------------------------------------------------------
def precompute_eq(g: Sequence[int]) -> List[int]:
    dim = len(g)
    dp = [0] * (1 << dim)
    dp[0] = (1 - g[0]) % P
    dp[1] = g[0] % P
    for i in range(1, dim):
        for b in range(1 << i):
            prev = dp[b]
            dp[b + (1 << i)] = prev * g[i] % P
            dp[b] = (prev - dp[b + (1 << i)]) % P
    return dp


def sparse_fix_variables(evals: Dict[int, int], num_vars: int, point: Sequence[int]):
    """SparseMultilinearExtension::fix_variables: windowed eq-table fold, LSB-first.
    Returns (dict, num_vars - len(point)).  Used at reference
    src/gkr_round_sumcheck/mod.rs:31 and :62."""
    dim_total = len(point)
    assert dim_total <= num_vars
    nnz = len(evals)
    window = max(nnz - 1, 0).bit_length() if nnz > 1 else 0  # ark_std::log2 = ceil(log2(n))
    last = dict(evals)
    pt = list(point)
    while pt:
        focus_len = window if (window > 0 and len(pt) > window) else len(pt)
        focus, pt = pt[:focus_len], pt[focus_len:]
        pre = precompute_eq(focus)
        res: Dict[int, int] = {}
        for old_idx, v in last.items():
            gz = pre[old_idx & ((1 << focus_len) - 1)]
            new_idx = old_idx >> focus_len
            res[new_idx] = (res.get(new_idx, 0) + gz * v) % P
        last = res
    return last, num_vars - dim_total


def sparse_to_dense(evals: Dict[int, int], num_vars: int) -> List[int]:
    out = [0] * (1 << num_vars)
    for i, v in evals.items():
        out[i] = v % P
    return out


def sparse_evaluate(evals: Dict[int, int], num_vars: int, point: Sequence[int]) -> int:
    assert len(point) == num_vars
    d, nv = sparse_fix_variables(evals, num_vars, point)
    assert nv == 0
    return d.get(0, 0)


# ---------------------------------------------------------------------------------------------
# GKR round sumcheck (reference src/gkr_round_sumcheck/mod.rs)
# ---------------------------------------------------------------------------------------------
def initialize_phase_one(f1: Dict[int, int], f1_nv: int, f3: Sequence[int], g: Sequence[int]):
    """gkr_round_sumcheck/mod.rs:22-42 -> (h_g dense list, f1_at_g dict)"""
    dim = (len(f3) - 1).bit_length()
    assert f1_nv == 3 * dim and len(g) == dim
    a_hg = [0] * (1 << dim)
    f1_at_g, _ = sparse_fix_variables(f1, f1_nv, g)
    for xy, v in f1_at_g.items():
        if v != 0:
            x = xy & ((1 << dim) - 1)
            y = xy >> dim
            a_hg[x] = (a_hg[x] + v * f3[y]) % P
    return a_hg, f1_at_g


def initialize_phase_two(f1_g: Dict[int, int], dim: int, u: Sequence[int]) -> List[int]:
    """gkr_round_sumcheck/mod.rs:57-63"""
    assert len(u) == dim
    d, nv = sparse_fix_variables(f1_g, 2 * dim, u)
    return sparse_to_dense(d, nv)


def gkr_prove(rng: Blake2b512Rng, f1: Dict[int, int], f2: Sequence[int], f3: Sequence[int],
              g: Sequence[int]):
    """gkr_round_sumcheck/mod.rs:93-139 -> (phase1 msgs, phase2 msgs, u, v)"""
    dim = (len(f2) - 1).bit_length()
    h_g, f1_g = initialize_phase_one(f1, 3 * dim, f3, g)
    poly = ListOfProductsOfPolynomials(dim)  # start_phase1_sumcheck, mod.rs:45-54
    poly.add_product([list(h_g), list(f2)], 1)
    st = prover_init(poly)
    vm, msgs1, u = None, [], []
    for _ in range(dim):
        pm = prove_round(st, vm)
        rng.feed_bytes(ser_prover_msg(pm))
        msgs1.append(pm)
        vm = sample_fr(rng)
        u.append(vm)
    f1_gu = initialize_phase_two(f1_g, dim, u)
    f2_u = dense_evaluate(f2, u)  # mod.rs:122
    f3_f2u = [f2_u * x % P for x in f3]  # start_phase2_sumcheck, mod.rs:66-82
    poly2 = ListOfProductsOfPolynomials(dim)
    poly2.add_product([f1_gu, f3_f2u], 1)
    st2 = prover_init(poly2)
    vm, msgs2, v = None, [], []
    for _ in range(dim):
        pm = prove_round(st2, vm)
        rng.feed_bytes(ser_prover_msg(pm))
        msgs2.append(pm)
        vm = sample_fr(rng)
        v.append(vm)
    return msgs1, msgs2, u, v


def gkr_verify(rng: Blake2b512Rng, dim: int, msgs1, msgs2, claimed_sum: int):
    """gkr_round_sumcheck/mod.rs:147-192 -> (u, v, expected_evaluation)"""
    rs = []
    for i in range(dim):
        rng.feed_bytes(ser_prover_msg(msgs1[i]))
        rs.append(sample_fr(rng))
    u, exp1 = check_and_generate_subclaim(dim, 2, msgs1, rs, claimed_sum)
    rs = []
    for i in range(dim):
        rng.feed_bytes(ser_prover_msg(msgs2[i]))
        rs.append(sample_fr(rng))
    v, exp2 = check_and_generate_subclaim(dim, 2, msgs2, rs, exp1)
    return u, v, exp2


def gkr_verify_subclaim(f1: Dict[int, int], f2, f3, g, u, v, expected: int) -> bool:
    """gkr_round_sumcheck/data_structures.rs:33-56"""
    dim = len(u)
    guv = list(g) + list(u) + list(v)
    actual = sparse_evaluate(f1, 3 * dim, guv) * dense_evaluate(f2, u) % P * dense_evaluate(f3, v) % P
    return actual == expected % P


# ---------------------------------------------------------------------------------------------
# Deterministic synthetic inputs: SplitMix64 keyed by (seed, stream, index)
# ---------------------------------------------------------------------------------------------
SEED = 0x5C20241008


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def synth_mont_limbs(seed: int, stream: int, index: int) -> Tuple[int, int, int, int]:
    """The synthetic table generator shared (by specification, not by code) with the HIP
    generator kernel and oracle.c: key = splitmix64(seed ^ stream*0xD1342543DE82EF95); limb k of attempt a is
    splitmix64(key ^ (splitmix64(4*index + k) + a*0x9E3779B97F4A7C15)); top bit cleared;
    first attempt with value < P wins.  Limbs are taken directly as Montgomery form, the same
    shape as ark-ff's sampler."""
    key = _splitmix64((seed ^ (stream * 0xD1342543DE82EF95)) & MASK64)
    attempt = 0
    while True:
        limbs = []
        for k in range(4):
            x = (_splitmix64((index * 4 + k) & MASK64) + attempt * 0x9E3779B97F4A7C15) & MASK64
            limbs.append(_splitmix64(key ^ x))
        limbs[3] &= MASK64 >> 1
        m = sum(limbs[i] << (64 * i) for i in range(4))
        if m < P:
            return tuple(limbs)  # type: ignore[return-value]
        attempt += 1


def synth_table(seed: int, stream: int, n: int) -> List[int]:
    """n canonical field elements of synthetic stream `stream`."""
    return [from_mont_limbs(synth_mont_limbs(seed, stream, i)) for i in range(n)]
