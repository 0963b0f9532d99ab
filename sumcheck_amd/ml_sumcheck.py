"""Host-side mirror of the reference's MLSumcheck API over the C ABI (include/sumcheck_hip.h).

Names, argument meaning and error behaviour follow the reference so that tests read like its own:
  ListOfProductsOfPolynomials / PolynomialInfo   reference src/ml_sumcheck/data_structures.rs:25-110
  ProverState / ProverMsg / IPForMLSumcheck      reference src/ml_sumcheck/protocol/prover.rs:13-153
  VerifierMsg / SubClaim / sample_round          reference src/ml_sumcheck/protocol/verifier.rs
  MLSumcheck::{prove, prove_as_subprotocol, verify, verify_as_subprotocol, extract_sum}
                                                 reference src/ml_sumcheck/mod.rs:24-101
  Blake2b512Rng / FeedableRNG                    reference src/rng.rs:11-81
The reference's panics surface as SumcheckError carrying the same message text.

All prove_round work happens in libsumcheck_hip.so on the GPU; this file only marshals.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import field
from ._lib import (SC_TABLES_BORROW, SC_TABLES_ON_DEVICE, PolyDesc, SumcheckError, check, lib)


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _np64(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class DenseMultilinearExtension:
    """ark_poly::DenseMultilinearExtension: `evaluations` is (2^num_vars, 4) uint64 Montgomery limbs, either a
    numpy array (host) or a torch int64 tensor on the GPU (HBM-resident)."""

    def __init__(self, num_vars: int, evaluations):
        self.num_vars = num_vars
        if _is_torch(evaluations):
            assert evaluations.is_contiguous() and evaluations.numel() == 4 << num_vars
            self.evaluations = evaluations
            self.on_device = evaluations.is_cuda
            if not self.on_device:
                self.evaluations = evaluations.numpy().view(np.uint64).reshape(-1, 4)
        else:
            self.evaluations = _np64(evaluations).reshape(-1, 4)
            assert self.evaluations.shape[0] == 1 << num_vars
            self.on_device = False

    @classmethod
    def from_evaluations_vec(cls, num_vars: int, evaluations):
        return cls(num_vars, evaluations)

    def data_ptr(self) -> int:
        return self.evaluations.data_ptr() if self.on_device else self.evaluations.ctypes.data

    def fix_variables(self, partial_point) -> "DenseMultilinearExtension":
        pt = _np64(partial_point).reshape(-1, 4)
        k = pt.shape[0]
        if self.on_device:
            import torch
            out = torch.empty((1 << (self.num_vars - k), 4), dtype=torch.int64, device=self.evaluations.device)
            check(lib().sc_fix_variables(C.c_void_p(self.data_ptr()), self.num_vars, _ptr(pt), k, C.c_void_p(out.data_ptr()),
                                         SC_TABLES_ON_DEVICE))
        else:
            out = np.empty((1 << (self.num_vars - k), 4), dtype=np.uint64)
            check(lib().sc_fix_variables(_ptr(self.evaluations), self.num_vars, _ptr(pt), k, _ptr(out), 0))
        return DenseMultilinearExtension(self.num_vars - k, out)

    def evaluate(self, point) -> np.ndarray:
        pt = _np64(point).reshape(-1, 4)
        assert pt.shape[0] == self.num_vars
        r = self.fix_variables(pt).evaluations
        if _is_torch(r):
            r = r.cpu().numpy().view(np.uint64)
        return np.asarray(r).reshape(4).copy()


@dataclass
class PolynomialInfo:  # data_structures.rs:47-55
    max_multiplicands: int
    num_variables: int


class ListOfProductsOfPolynomials:
    """data_structures.rs:25-110.  De-duplication of multiplicands is by object identity, the analogue of the
    reference's Rc pointer lookup (data_structures.rs:85-93)."""

    def __init__(self, num_variables: int):
        self.max_multiplicands = 0
        self.num_variables = num_variables
        self.products: List[Tuple[np.ndarray, List[int]]] = []
        self.flattened_ml_extensions: List[DenseMultilinearExtension] = []
        self._lookup = {}

    @classmethod
    def new(cls, num_variables: int) -> "ListOfProductsOfPolynomials":
        return cls(num_variables)

    def add_product(self, product: Iterable[DenseMultilinearExtension], coefficient) -> None:
        product = list(product)
        assert len(product) > 0  # data_structures.rs:78
        self.max_multiplicands = max(self.max_multiplicands, len(product))
        indexed = []
        for m in product:
            assert m.num_vars == self.num_variables, "product has a multiplicand with wrong number of variables"
            key = id(m)
            if key in self._lookup:
                indexed.append(self._lookup[key])
            else:
                idx = len(self.flattened_ml_extensions)
                self.flattened_ml_extensions.append(m)
                self._lookup[key] = idx
                indexed.append(idx)
        self.products.append((_np64(coefficient).reshape(4).copy(), indexed))

    def info(self) -> PolynomialInfo:
        return PolynomialInfo(self.max_multiplicands, self.num_variables)

    def evaluate(self, point) -> np.ndarray:
        """data_structures.rs:99-109: sum_k c_k prod_j T_j(point) -- one library call (sc_poly_evaluate: all tables folded on
        the GPU three variables per pass, the K + sum m_k scalar products on the library's host side)."""
        return self.evaluate_with_tables(point)[0]

    def evaluate_with_tables(self, point):
        """-> (value (4,), per-table evaluations (U, 4))"""
        pt = np.ascontiguousarray(_np64(point).reshape(-1, 4))
        assert pt.shape[0] == self.num_variables
        d, keep = self._desc()
        out = np.zeros(4, dtype=np.uint64)
        tv = np.zeros((max(len(self.flattened_ml_extensions), 1), 4), dtype=np.uint64)
        check(lib().sc_poly_evaluate(C.byref(d), _ptr(pt) if pt.size else None, _ptr(out), _ptr(tv)))
        del keep
        return out, tv[: len(self.flattened_ml_extensions)]

    # ---- marshalling -----------------------------------------------------------------------------
    def _desc(self, borrow: bool = False):
        K = len(self.products)
        coeffs = np.ascontiguousarray(np.stack([c for c, _ in self.products])) if K else np.zeros((1, 4), np.uint64)
        offs, idx = [0], []
        for _, ix in self.products:
            idx.extend(ix)
            offs.append(len(idx))
        offsets = np.asarray(offs, dtype=np.uint32)
        indices = np.asarray(idx if idx else [0], dtype=np.uint32)
        U = len(self.flattened_ml_extensions)
        dev = [t.on_device for t in self.flattened_ml_extensions]
        if any(dev) and not all(dev):
            raise ValueError("mixing host and device tables in one polynomial is not supported")
        tabs = (C.c_void_p * max(U, 1))()
        for i, t in enumerate(self.flattened_ml_extensions):
            tabs[i] = t.data_ptr()
        d = PolyDesc()
        d.num_vars = self.num_variables
        d.max_multiplicands = self.max_multiplicands
        d.n_products = K
        d.coeffs = coeffs.ctypes.data_as(C.POINTER(C.c_uint64))
        d.prod_offsets = offsets.ctypes.data_as(C.POINTER(C.c_uint32))
        d.prod_indices = indices.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_tables = U
        d.tables = C.cast(tabs, C.POINTER(C.c_void_p))
        d.flags = (SC_TABLES_ON_DEVICE if (U and all(dev)) else 0) | (SC_TABLES_BORROW if borrow else 0)
        keep = (coeffs, offsets, indices, tabs)
        return d, keep


@dataclass
class ProverMsg:  # prover.rs:13-17
    evaluations: np.ndarray  # (deg+1, 4)


@dataclass
class VerifierMsg:  # verifier.rs:10-15
    randomness: np.ndarray  # (4,)


@dataclass
class SubClaim:  # verifier.rs:29-34
    point: np.ndarray  # (nv, 4)
    expected_evaluation: np.ndarray  # (4,)


class ProverState:
    """prover.rs:19-33 with flattened_ml_extensions resident in HBM behind an sc_prover handle."""

    def __init__(self, handle: C.c_void_p, poly: ListOfProductsOfPolynomials):
        self._h = handle
        self.list_of_products = [(c.copy(), list(ix)) for c, ix in poly.products]
        self.num_vars = poly.num_variables
        self.max_multiplicands = poly.max_multiplicands
        self._n_tables = len(poly.flattened_ml_extensions)

    @property
    def round(self) -> int:
        r = C.c_uint32()
        check(lib().sc_prover_state(self._h, None, None, None, C.byref(r)))
        return r.value

    @property
    def randomness(self) -> np.ndarray:
        buf = np.zeros((self.num_vars + 1, 4), dtype=np.uint64)
        n = C.c_uint32()
        check(lib().sc_prover_state(self._h, _ptr(buf), C.byref(n), None, None))
        return buf[: n.value].copy()

    @property
    def flattened_ml_extensions(self) -> List[DenseMultilinearExtension]:
        rnd = self.round
        nvars = self.num_vars - max(rnd - 1, 0)
        buf = np.zeros((self._n_tables, 1 << nvars, 4), dtype=np.uint64)
        check(lib().sc_prover_state(self._h, None, None, _ptr(buf), None))
        return [DenseMultilinearExtension(nvars, buf[u]) for u in range(self._n_tables)]

    def reset(self) -> None:
        """rewind a borrowing handle to round 0 over the same resident tables (no allocation)"""
        check(lib().sc_prover_reset(self._h, None, 0))

    def set_timing(self, on: bool = True) -> None:
        check(lib().sc_prover_set_timing(self._h, 1 if on else 0))

    def get_timing(self):
        """-> (ms per product, launches per product, accumulated per-round span in ms)"""
        K = len(self.list_of_products)
        ms = (C.c_double * K)()
        ln = (C.c_uint64 * K)()
        tot = C.c_double()
        check(lib().sc_prover_get_timing(self._h, ms, ln, C.byref(tot)))
        return list(ms), list(ln), tot.value

    def prove(self, fs_rng: Optional["Blake2b512Rng"] = None) -> np.ndarray:
        """MLSumcheck::prove_as_subprotocol's loop on this handle (must be at round 0) -> (nv, deg+1, 4)"""
        proof = np.empty((self.num_vars, self.max_multiplicands + 1, 4), dtype=np.uint64)
        check(lib().sc_ml_prove_handle(self._h, fs_rng._h if fs_rng is not None else None, _ptr(proof)))
        return proof

    def last_round_ms(self) -> float:
        ms = C.c_float()
        check(lib().sc_prover_last_round_ms(self._h, C.byref(ms)))
        return ms.value

    def close(self):
        if self._h:
            lib().sc_prover_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Blake2b512Rng:
    """FeedableRNG + RngCore of reference src/rng.rs"""

    def __init__(self):
        self._h = C.c_void_p(lib().sc_rng_setup())

    @classmethod
    def setup(cls) -> "Blake2b512Rng":
        return cls()

    def feed(self, msg) -> None:
        """feed(&M) with M: bytes (serialised as-is, e.g. b"Test Trivial Works"), ProverMsg, or PolynomialInfo."""
        if isinstance(msg, (bytes, bytearray)):
            lib().sc_rng_feed_bytes(self._h, bytes(msg), len(msg))
        elif isinstance(msg, ProverMsg):
            ev = _np64(msg.evaluations)
            lib().sc_rng_feed_prover_msg(self._h, _ptr(ev), ev.shape[0])
        elif isinstance(msg, PolynomialInfo):
            lib().sc_rng_feed_poly_info(self._h, msg.max_multiplicands, msg.num_variables)
        else:
            raise TypeError(f"cannot serialise {type(msg)}")

    def fill_bytes(self, n: int) -> bytes:
        buf = (C.c_uint8 * max(n, 1))()
        lib().sc_rng_fill_bytes(self._h, C.cast(buf, C.c_void_p), n)
        return bytes(buf[:n])

    def next_u64(self) -> int:
        return int.from_bytes(self.fill_bytes(8), "little")

    def next_u32(self) -> int:
        return int.from_bytes(self.fill_bytes(4), "little")

    def sample_fr(self) -> np.ndarray:
        out = np.empty(4, dtype=np.uint64)
        lib().sc_rng_sample_fr(self._h, _ptr(out))
        return out

    def __del__(self):
        try:
            if self._h:
                lib().sc_rng_free(self._h)
                self._h = None
        except Exception:
            pass


class IPForMLSumcheck:
    @staticmethod
    def prover_init(polynomial: ListOfProductsOfPolynomials, borrow: bool = False, streamed_chunk_log2: Optional[int] = None) -> ProverState:
        """prover.rs:49-69.  `borrow=True` (device tables only) skips the deep copy.  `streamed_chunk_log2` (host tables only; 0 = the
        library's default chunk): out-of-core mode -- the tables stay in host memory and are streamed through HBM in rounds 1 and 2
        (sc_prover_init_streamed); the caller keeps them alive and unchanged until round 2 has returned."""
        d, keep = polynomial._desc(borrow)
        h = C.c_void_p()
        if streamed_chunk_log2 is not None:
            check(lib().sc_prover_init_streamed(C.byref(d), int(streamed_chunk_log2), C.byref(h)))
            st = ProverState(h, polynomial)
            st._keep = (keep, list(polynomial.flattened_ml_extensions))  # the streamed host tables must outlive round 2
            return st
        check(lib().sc_prover_init(C.byref(d), C.byref(h)))
        del keep
        return ProverState(h, polynomial)

    @staticmethod
    def prove_round(prover_state: ProverState, v_msg: Optional[VerifierMsg]) -> ProverMsg:
        """prover.rs:74-153"""
        out = np.empty((prover_state.max_multiplicands + 1, 4), dtype=np.uint64)
        r = _np64(v_msg.randomness).reshape(4) if v_msg is not None else None
        check(lib().sc_prove_round(prover_state._h, _ptr(r) if r is not None else None, _ptr(out)))
        return ProverMsg(out)

    @staticmethod
    def sample_round(rng) -> VerifierMsg:
        """verifier.rs:128-131"""
        return VerifierMsg(rng.sample_fr())


class MLSumcheck:
    @staticmethod
    def extract_sum(proof: Sequence[ProverMsg]) -> np.ndarray:  # mod.rs:26-28
        return field.add(proof[0].evaluations[0], proof[0].evaluations[1])

    @staticmethod
    def prove(polynomial: ListOfProductsOfPolynomials) -> List[ProverMsg]:  # mod.rs:42-45
        """No prover state is returned, so the library reads device-resident tables in place and keeps the prover it built for the
        next proof of the same shape (sc_ml_prove with a null state pointer)."""
        d, keep = polynomial._desc(False)
        D = polynomial.max_multiplicands + 1
        proof = np.empty((max(polynomial.num_variables, 1), D, 4), dtype=np.uint64)
        check(lib().sc_ml_prove(C.byref(d), None, _ptr(proof), None))  # null rng = a fresh Blake2b512Rng::setup()
        del keep
        return [ProverMsg(proof[i].copy()) for i in range(polynomial.num_variables)]

    @staticmethod
    def prove_as_subprotocol(fs_rng: Blake2b512Rng, polynomial: ListOfProductsOfPolynomials, borrow: bool = False):
        """mod.rs:50-70: the whole Fiat-Shamir loop runs inside the library (sc_ml_prove)."""
        d, keep = polynomial._desc(borrow)
        D = polynomial.max_multiplicands + 1
        proof = np.empty((max(polynomial.num_variables, 1), D, 4), dtype=np.uint64)
        h = C.c_void_p()
        check(lib().sc_ml_prove(C.byref(d), fs_rng._h, _ptr(proof), C.byref(h)))
        del keep
        msgs = [ProverMsg(proof[i].copy()) for i in range(polynomial.num_variables)]
        return msgs, ProverState(h, polynomial)

    @staticmethod
    def verify(polynomial_info: PolynomialInfo, claimed_sum, proof: Sequence[ProverMsg]) -> SubClaim:  # mod.rs:73-80
        return MLSumcheck.verify_as_subprotocol(Blake2b512Rng.setup(), polynomial_info, claimed_sum, proof)

    @staticmethod
    def verify_as_subprotocol(fs_rng: Blake2b512Rng, polynomial_info: PolynomialInfo, claimed_sum,
                              proof: Sequence[ProverMsg]) -> SubClaim:
        """mod.rs:84-100; rejects with SumcheckError(SC_ERR_REJECT, "Prover message is not consistent with the claim.")"""
        nv = polynomial_info.num_variables
        D = polynomial_info.max_multiplicands + 1
        if len(proof) < nv:
            raise SumcheckError(5, "proof is incomplete")
        msgs = [_np64(m.evaluations).reshape(-1, 4) for m in proof[:nv]]
        for m in msgs:  # verifier.rs:60-62 panics on a message of the wrong length
            if m.shape != (D, 4):
                raise SumcheckError(5, "incorrect number of evaluations")
        flat = np.ascontiguousarray(np.stack(msgs)) if nv else np.zeros((1, D, 4), np.uint64)
        point = np.empty((max(nv, 1), 4), dtype=np.uint64)
        exp = np.empty(4, dtype=np.uint64)
        cs = _np64(claimed_sum).reshape(4)  # bound to a local: the converted array must outlive the call
        check(lib().sc_ml_verify(nv, polynomial_info.max_multiplicands, _ptr(cs), _ptr(flat), nv * D, fs_rng._h, _ptr(point), _ptr(exp)))
        return SubClaim(point[:nv].copy(), exp)


def interpolate_uni_poly(p_i, eval_at) -> np.ndarray:
    """verifier.rs:139-251"""
    p_i = _np64(p_i).reshape(-1, 4)
    out = np.empty(4, dtype=np.uint64)
    at = _np64(eval_at).reshape(4)
    check(lib().sc_interpolate_uni_poly(_ptr(p_i), p_i.shape[0], _ptr(at), _ptr(out)))
    return out
