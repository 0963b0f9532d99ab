"""ctypes binding to oracle/liboracle.so (the C restatement in oracle/oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only.

Field elements travel as numpy uint64 arrays with a trailing axis of 4 (little-endian limbs,
Montgomery form) -- the in-memory layout of ark_ff::Fp<MontBackend<FrConfig,4>,4>.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_lib = None

P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % P
RINV = pow(R, -1, P)

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_rng_setup.restype = C.c_void_p
        _lib.orc_sparse_fix_variables.restype = C.c_uint64
        _lib.orc_gkr_phase_one.restype = C.c_uint64
        _lib.orc_prover_state.restype = C.c_uint32
    return _lib


def _p64(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _p32(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


# ---- int <-> limb helpers -----------------------------------------------------------------------
def ints_to_mont(vals: Sequence[int]) -> np.ndarray:
    """canonical python ints -> (n,4) uint64 Montgomery limbs"""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = (int(v) % P) * R % P
        for k in range(4):
            out[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def mont_to_ints(a: np.ndarray) -> List[int]:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    res = []
    for row in a:
        m = int(row[0]) | (int(row[1]) << 64) | (int(row[2]) << 128) | (int(row[3]) << 192)
        res.append(m * RINV % P)
    return res


def synth_table(seed: int, stream: int, n: int, first: int = 0) -> np.ndarray:
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_synth_table(C.c_uint64(seed), C.c_uint64(stream), C.c_uint64(first), C.c_uint64(n), _p64(out))
    return out


def fr_binop(name: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.empty(4, dtype=np.uint64)
    getattr(lib(), "orc_fr_" + name)(_p64(np.ascontiguousarray(a)), _p64(np.ascontiguousarray(b)), _p64(out))
    return out


def fix_variables(table: np.ndarray, point: np.ndarray) -> np.ndarray:
    table = np.ascontiguousarray(table, dtype=np.uint64)
    point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
    n = table.shape[0]
    nv = n.bit_length() - 1
    k = point.shape[0]
    out = np.empty((n >> k, 4), dtype=np.uint64)
    lib().orc_fix_variables(_p64(table), C.c_uint32(nv), _p64(point), C.c_uint32(k), _p64(out))
    return out


class PolyDesc:
    """Flattened ListOfProductsOfPolynomials: coeffs (K,4), CSR offsets/indices, tables list."""

    def __init__(self, num_vars: int, products: Sequence[Tuple[np.ndarray, Sequence[int]]], tables: Sequence[np.ndarray]):
        self.num_vars = num_vars
        self.coeffs = np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.uint64).reshape(4) for c, _ in products])
                                           if products else np.zeros((0, 4), np.uint64))
        offs, idx = [0], []
        for _, ix in products:
            idx.extend(int(i) for i in ix)
            offs.append(len(idx))
        self.offsets = np.asarray(offs, dtype=np.uint32)
        self.indices = np.asarray(idx, dtype=np.uint32)
        self.max_multiplicands = max((len(ix) for _, ix in products), default=0)
        self.tables = [np.ascontiguousarray(t, dtype=np.uint64) for t in tables]
        self.n_products = len(products)

    def table_ptrs(self):
        arr = (u64p * len(self.tables))()
        for i, t in enumerate(self.tables):
            arr[i] = _p64(t)
        return arr

    def field_ops(self) -> int:
        """mul+add+sub as executed by the reference algorithm."""
        nv, D = self.num_vars, self.max_multiplicands + 1
        ms = np.diff(self.offsets.astype(np.int64))
        ops_sum = ((1 << nv) - 1) * int(sum(2 * m * D + m + D for m in ms))
        ops_fix = 3 * len(self.tables) * ((1 << nv) - 2)
        return ops_sum + ops_fix

    def algorithmic_bytes(self) -> int:
        return 32 * len(self.tables) * (4 * (1 << self.num_vars) - 6)


class Prover:
    def __init__(self, desc: PolyDesc, threads: int = 1, improved_fix: bool = False):
        self.desc = desc
        self._h = C.c_void_p()
        rc = lib().orc_prover_init(C.c_uint32(desc.num_vars), C.c_uint32(desc.max_multiplicands), C.c_uint32(desc.n_products),
                                   _p64(desc.coeffs), _p32(desc.offsets), _p32(desc.indices), C.c_uint32(len(desc.tables)),
                                   desc.table_ptrs(), C.c_int(threads), C.byref(self._h))
        if rc != 0:
            raise RuntimeError({1: "Attempt to prove a constant."}.get(rc, f"oracle error {rc}"))
        if improved_fix:
            lib().orc_prover_set_improved_fix(self._h, 1)
        self.D = desc.max_multiplicands + 1

    def prove_round(self, r: Optional[np.ndarray]) -> np.ndarray:
        out = np.empty((self.D, 4), dtype=np.uint64)
        rp = _p64(np.ascontiguousarray(r, dtype=np.uint64)) if r is not None else None
        rc = lib().orc_prove_round(self._h, rp, _p64(out))
        if rc != 0:
            raise RuntimeError({2: "first round should be prover first.", 3: "verifier message is empty",
                                4: "Prover is not active"}.get(rc, f"oracle error {rc}"))
        return out

    def state(self):
        nv = self.desc.num_vars
        rnd = C.c_uint32()
        rand = np.zeros((nv, 4), dtype=np.uint64)
        n_rand = lib().orc_prover_state(self._h, _p64(rand), None, C.byref(rnd))
        bound = max(rnd.value - 1, 0)
        tabs = np.zeros((len(self.desc.tables), 1 << (nv - bound), 4), dtype=np.uint64)
        lib().orc_prover_state(self._h, None, _p64(tabs), None)
        return rand[:n_rand], tabs, rnd.value

    def push_randomness(self, r: np.ndarray):
        lib().orc_prover_push_randomness(self._h, _p64(np.ascontiguousarray(r, dtype=np.uint64)))

    def times(self):
        """seconds so far in (prover_init's deep copy, fix_variables, the sums): bench.py's per-phase split"""
        t = (C.c_double * 3)()
        lib().orc_prover_times(self._h, t)
        return tuple(t)

    def close(self):
        if self._h:
            lib().orc_prover_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Rng:
    """Blake2b512Rng (reference src/rng.rs:22-81)"""

    def __init__(self):
        self._h = C.c_void_p(lib().orc_rng_setup())

    def feed_bytes(self, b: bytes):
        buf = (C.c_uint8 * len(b)).from_buffer_copy(b) if len(b) else (C.c_uint8 * 1)()
        lib().orc_rng_feed_bytes(self._h, buf, C.c_size_t(len(b)))

    def fill_bytes(self, n: int) -> bytes:
        buf = (C.c_uint8 * max(n, 1))()
        lib().orc_rng_fill_bytes(self._h, buf, C.c_size_t(n))
        return bytes(buf[:n])

    def sample_fr(self) -> np.ndarray:
        out = np.empty(4, dtype=np.uint64)
        lib().orc_rng_sample_fr(self._h, _p64(out))
        return out

    def feed_prover_msg(self, evals: np.ndarray):
        evals = np.ascontiguousarray(evals, dtype=np.uint64)
        lib().orc_rng_feed_prover_msg(self._h, _p64(evals), C.c_uint32(evals.shape[0]))

    def feed_poly_info(self, max_multiplicands: int, num_variables: int):
        lib().orc_rng_feed_poly_info(self._h, C.c_uint64(max_multiplicands), C.c_uint64(num_variables))

    def __del__(self):
        try:
            lib().orc_rng_free(self._h)
        except Exception:
            pass


def blake2b512(b: bytes) -> bytes:
    out = (C.c_uint8 * 64)()
    buf = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")
    lib().orc_blake2b512(buf, C.c_size_t(len(b)), out)
    return bytes(out)


def ml_prove(desc: PolyDesc, rng: Optional[Rng] = None, threads: int = 1):
    """-> (proof (nv,D,4), randomness (nv,4))"""
    rng = rng or Rng()
    D = desc.max_multiplicands + 1
    proof = np.empty((desc.num_vars, D, 4), dtype=np.uint64)
    rand = np.empty((desc.num_vars, 4), dtype=np.uint64)
    rc = lib().orc_ml_prove(rng._h, C.c_uint32(desc.num_vars), C.c_uint32(desc.max_multiplicands), C.c_uint32(desc.n_products),
                            _p64(desc.coeffs), _p32(desc.offsets), _p32(desc.indices), C.c_uint32(len(desc.tables)),
                            desc.table_ptrs(), C.c_int(threads), _p64(proof), _p64(rand))
    if rc != 0:
        raise RuntimeError({1: "Attempt to prove a constant."}.get(rc, f"oracle error {rc}"))
    return proof, rand


def last_prove_times():
    """(init copy, fix_variables, sums) seconds of the last ml_prove of this process"""
    t = (C.c_double * 3)()
    lib().orc_last_prove_times(t)
    return tuple(t)


def check_and_generate_subclaim(nv: int, max_mult: int, polys: np.ndarray, randomness: np.ndarray, asserted_sum: np.ndarray):
    """-> (accepted: bool, expected_evaluation (4,))"""
    out = np.zeros(4, dtype=np.uint64)
    rc = lib().orc_check_and_generate_subclaim(C.c_uint32(nv), C.c_uint32(max_mult), _p64(np.ascontiguousarray(polys, dtype=np.uint64)),
                                               _p64(np.ascontiguousarray(randomness, dtype=np.uint64)),
                                               _p64(np.ascontiguousarray(asserted_sum, dtype=np.uint64)), _p64(out))
    return rc == 0, out


def ml_verify(desc_info: Tuple[int, int], claimed_sum: np.ndarray, proof: np.ndarray, rng: Optional[Rng] = None):
    """MLSumcheck::verify_as_subprotocol (reference src/ml_sumcheck/mod.rs:84-100)
    -> (accepted, point (nv,4), expected (4,))"""
    max_mult, nv = desc_info
    rng = rng or Rng()
    rng.feed_poly_info(max_mult, nv)
    rs = np.empty((nv, 4), dtype=np.uint64)
    for i in range(nv):
        rng.feed_prover_msg(proof[i])
        rs[i] = rng.sample_fr()
    ok, exp = check_and_generate_subclaim(nv, max_mult, proof, rs, claimed_sum)
    return ok, rs, exp


def poly_evaluate(desc: PolyDesc, point: np.ndarray) -> np.ndarray:
    out = np.empty(4, dtype=np.uint64)
    lib().orc_poly_evaluate(C.c_uint32(desc.num_vars), C.c_uint32(desc.n_products), _p64(desc.coeffs), _p32(desc.offsets),
                            _p32(desc.indices), C.c_uint32(len(desc.tables)), desc.table_ptrs(),
                            _p64(np.ascontiguousarray(point, dtype=np.uint64)), _p64(out))
    return out


def interpolate_uni_poly(p_i: np.ndarray, eval_at: np.ndarray) -> np.ndarray:
    out = np.empty(4, dtype=np.uint64)
    p_i = np.ascontiguousarray(p_i, dtype=np.uint64)
    lib().orc_interpolate_uni_poly(_p64(p_i), C.c_uint32(p_i.shape[0]), _p64(np.ascontiguousarray(eval_at, dtype=np.uint64)), _p64(out))
    return out


def sparse_fix_variables(idx: np.ndarray, vals: np.ndarray, point: np.ndarray):
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    vals = np.ascontiguousarray(vals, dtype=np.uint64)
    point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
    n = idx.shape[0]
    oi = np.empty(max(n, 1), dtype=np.uint64)
    ov = np.empty((max(n, 1), 4), dtype=np.uint64)
    m = lib().orc_sparse_fix_variables(_p64(idx), _p64(vals), C.c_uint64(n), _p64(point), C.c_uint32(point.shape[0]), _p64(oi), _p64(ov))
    return oi[:m].copy(), ov[:m].copy()


def gkr_phase_one(idx, vals, dim: int, f3, g):
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    vals = np.ascontiguousarray(vals, dtype=np.uint64)
    n = idx.shape[0]
    h_g = np.empty((1 << dim, 4), dtype=np.uint64)
    oi = np.empty(max(n, 1), dtype=np.uint64)
    ov = np.empty((max(n, 1), 4), dtype=np.uint64)
    m = lib().orc_gkr_phase_one(_p64(idx), _p64(vals), C.c_uint64(n), C.c_uint32(dim), _p64(np.ascontiguousarray(f3, dtype=np.uint64)),
                                _p64(np.ascontiguousarray(g, dtype=np.uint64)), _p64(h_g), _p64(oi), _p64(ov))
    return h_g, oi[:m].copy(), ov[:m].copy()


def gkr_phase_two(f1g_idx, f1g_vals, dim: int, u):
    out = np.empty((1 << dim, 4), dtype=np.uint64)
    f1g_idx = np.ascontiguousarray(f1g_idx, dtype=np.uint64)
    lib().orc_gkr_phase_two(_p64(f1g_idx), _p64(np.ascontiguousarray(f1g_vals, dtype=np.uint64)), C.c_uint64(f1g_idx.shape[0]),
                            C.c_uint32(dim), _p64(np.ascontiguousarray(u, dtype=np.uint64)), _p64(out))
    return out


def gkr_prove(idx, vals, dim: int, f2, f3, g, rng: Optional[Rng] = None, threads: int = 1):
    """-> (proof (2,dim,3,4), uv (2,dim,4))"""
    rng = rng or Rng()
    idx = np.ascontiguousarray(idx, dtype=np.uint64)
    proof = np.empty((2, dim, 3, 4), dtype=np.uint64)
    uv = np.empty((2, dim, 4), dtype=np.uint64)
    rc = lib().orc_gkr_prove(rng._h, _p64(idx), _p64(np.ascontiguousarray(vals, dtype=np.uint64)), C.c_uint64(idx.shape[0]),
                             C.c_uint32(dim), _p64(np.ascontiguousarray(f2, dtype=np.uint64)),
                             _p64(np.ascontiguousarray(f3, dtype=np.uint64)), _p64(np.ascontiguousarray(g, dtype=np.uint64)),
                             C.c_int(threads), _p64(proof), _p64(uv))
    assert rc == 0
    return proof, uv


def cpu_quota_cores():
    """CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited.  The GPU boxes show all
    256 hardware threads of their host but run under a quota (16 cores when probed): more OpenMP threads than that are throttled, not run."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(round(int(q) / int(per))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, int(round(q / per)))
    except Exception:
        pass
    return None


def max_threads() -> int:
    """threads worth starting: what OpenMP would start, capped by the container's CPU quota"""
    n = int(lib().orc_max_threads())
    q = cpu_quota_cores()
    return min(n, q) if q else n
