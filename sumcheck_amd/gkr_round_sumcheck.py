"""Host-side mirror of the reference's GKR round sumcheck API over the C ABI.

  initialize_phase_one / start_phase1_sumcheck / initialize_phase_two / start_phase2_sumcheck
                                   reference src/gkr_round_sumcheck/mod.rs:22-82
  GKRRoundSumcheck::{prove, verify} reference src/gkr_round_sumcheck/mod.rs:84-193
  GKRProof / GKRRoundSumcheckSubClaim reference src/gkr_round_sumcheck/data_structures.rs:9-57

The prover side (sparse fold, scatter, both sumcheck phases) runs on the GPU inside libsumcheck_hip.so.
The verifier side is O(dim) scalar work and stays on the host, like the reference's.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

from . import field
from ._lib import SC_TABLES_ON_DEVICE, check, lib
from .ml_sumcheck import (Blake2b512Rng, DenseMultilinearExtension, IPForMLSumcheck, ListOfProductsOfPolynomials, ProverMsg,
                          ProverState, SumcheckError, _np64, _ptr, interpolate_uni_poly)


def _is_cuda(x) -> bool:
    return type(x).__module__.startswith("torch") and x.is_cuda


class SparseMultilinearExtension:
    """ark_poly::SparseMultilinearExtension: `indices` (nnz,) uint64 distinct, `values` (nnz, 4) Montgomery limbs -- numpy arrays
    (host) or torch int64 tensors on the GPU (HBM-resident; the library then reads them in place)."""

    def __init__(self, num_vars: int, indices, values):
        self.num_vars = num_vars
        self.on_device = _is_cuda(indices) and _is_cuda(values)
        if self.on_device:
            assert indices.is_contiguous() and values.is_contiguous() and values.numel() == 4 * indices.numel()
            self.indices, self.values = indices, values
        else:
            self.indices = np.ascontiguousarray(indices, dtype=np.uint64).reshape(-1)
            self.values = _np64(values).reshape(-1, 4)
            assert self.indices.shape[0] == self.values.shape[0]

    @property
    def nnz(self) -> int:
        return int(self.indices.shape[0])

    def _ptrs(self):
        if self.on_device:
            return C.c_void_p(self.indices.data_ptr()), C.c_void_p(self.values.data_ptr())
        return _ptr(self.indices), _ptr(self.values)

    def to_host(self) -> "SparseMultilinearExtension":
        if not self.on_device:
            return self
        return SparseMultilinearExtension(self.num_vars, self.indices.cpu().numpy().view(np.uint64), self.values.cpu().numpy().view(np.uint64))

    @classmethod
    def from_evaluations(cls, num_vars: int, evaluations):
        idx = [i for i, _ in evaluations]
        vals = [v for _, v in evaluations]
        return cls(num_vars, np.asarray(idx, dtype=np.uint64), np.stack(vals) if vals else np.zeros((0, 4), np.uint64))

    def evaluate(self, point) -> np.ndarray:
        """ark-poly SparseMultilinearExtension::evaluate on the GPU (sc_sparse_evaluate): the verifier's oracle query
        f1(g, u, v) of verify_subclaim (data_structures.rs:53)"""
        h = self.to_host()
        pt = np.ascontiguousarray(np.asarray(point, dtype=np.uint64).reshape(-1, 4))
        assert pt.shape[0] == self.num_vars
        out = np.zeros(4, dtype=np.uint64)
        check(lib().sc_sparse_evaluate(_ptr(h.indices), _ptr(h.values), h.indices.shape[0], self.num_vars,
                                       _ptr(pt) if pt.size else None, _ptr(out)))
        return out


def _dense_ptr(m: DenseMultilinearExtension):
    return C.c_void_p(m.data_ptr()) if m.on_device else _ptr(m.evaluations)


def initialize_phase_one(f1: SparseMultilinearExtension, f3: DenseMultilinearExtension, g):
    """mod.rs:22-42 -> (h_g dense, f1_at_g sparse).  Device-resident inputs (f1 and f3 both on the GPU) give device-resident
    outputs, produced in place by the library."""
    dim = f3.num_vars
    assert f1.num_vars == dim * 3
    g = _np64(g).reshape(-1, 4)
    assert g.shape[0] == dim
    nnz = f1.nnz
    dev = f1.on_device and f3.on_device
    assert dev or not (f1.on_device or f3.on_device), "mixing host and device inputs is not supported"
    n1 = C.c_uint64()
    i_ptr, v_ptr = f1._ptrs()
    if dev:
        import torch
        h_g = torch.empty((1 << dim, 4), dtype=torch.int64, device=f3.evaluations.device)
        oi = torch.empty(max(nnz, 1), dtype=torch.int64, device=h_g.device)
        ov = torch.empty((max(nnz, 1), 4), dtype=torch.int64, device=h_g.device)
        torch.cuda.current_stream(h_g.device).synchronize()  # the library works on its own stream
        check(lib().sc_gkr_phase_one(i_ptr, v_ptr, nnz, dim, _dense_ptr(f3), _ptr(g), SC_TABLES_ON_DEVICE, C.c_void_p(h_g.data_ptr()),
                                     C.c_void_p(oi.data_ptr()), C.c_void_p(ov.data_ptr()), C.byref(n1)))
        return DenseMultilinearExtension(dim, h_g), SparseMultilinearExtension(2 * dim, oi[: n1.value].contiguous(), ov[: n1.value].contiguous())
    h_g = np.empty((1 << dim, 4), dtype=np.uint64)
    oi = np.empty(max(nnz, 1), dtype=np.uint64)
    ov = np.empty((max(nnz, 1), 4), dtype=np.uint64)
    check(lib().sc_gkr_phase_one(i_ptr, v_ptr, nnz, dim, _dense_ptr(f3), _ptr(g), 0, _ptr(h_g), _ptr(oi), _ptr(ov), C.byref(n1)))
    return DenseMultilinearExtension(dim, h_g), SparseMultilinearExtension(2 * dim, oi[: n1.value].copy(), ov[: n1.value].copy())


def start_phase1_sumcheck(h_g: DenseMultilinearExtension, f2: DenseMultilinearExtension) -> ProverState:
    """mod.rs:45-54"""
    dim = h_g.num_vars
    assert f2.num_vars == dim
    poly = ListOfProductsOfPolynomials.new(dim)
    poly.add_product([h_g, f2], field.ONE)
    return IPForMLSumcheck.prover_init(poly)


def initialize_phase_two(f1_g: SparseMultilinearExtension, u) -> DenseMultilinearExtension:
    """mod.rs:57-63"""
    u = _np64(u).reshape(-1, 4)
    assert u.shape[0] * 2 == f1_g.num_vars
    dim = u.shape[0]
    i_ptr, v_ptr = f1_g._ptrs()
    if f1_g.on_device:
        import torch
        out = torch.empty((1 << dim, 4), dtype=torch.int64, device=f1_g.values.device)
        torch.cuda.current_stream(out.device).synchronize()
        check(lib().sc_gkr_phase_two(i_ptr, v_ptr, f1_g.nnz, dim, _ptr(u), SC_TABLES_ON_DEVICE, C.c_void_p(out.data_ptr())))
        return DenseMultilinearExtension(dim, out)
    out = np.empty((1 << dim, 4), dtype=np.uint64)
    check(lib().sc_gkr_phase_two(i_ptr, v_ptr, f1_g.nnz, dim, _ptr(u), 0, _ptr(out)))
    return DenseMultilinearExtension(dim, out)


def scale(m: DenseMultilinearExtension, scalar) -> DenseMultilinearExtension:
    """`DenseMultilinearExtension::zero() += (scalar, &m)` (mod.rs:71-75) on the GPU: every evaluation times one field element"""
    sv = _np64(scalar).reshape(4)
    n = 1 << m.num_vars
    if m.on_device:
        import torch
        out = torch.empty_like(m.evaluations)
        torch.cuda.current_stream(out.device).synchronize()
        check(lib().sc_dense_scale(_dense_ptr(m), n, _ptr(sv), C.c_void_p(out.data_ptr()), SC_TABLES_ON_DEVICE))
    else:
        out = np.empty((n, 4), dtype=np.uint64)
        check(lib().sc_dense_scale(_dense_ptr(m), n, _ptr(sv), _ptr(out), 0))
    return DenseMultilinearExtension(m.num_vars, out)


def start_phase2_sumcheck(f1_gu: DenseMultilinearExtension, f3: DenseMultilinearExtension, f2_u) -> ProverState:
    """mod.rs:66-82: f3 scaled by the scalar f2(u) (on the GPU), then one product of two tables"""
    dim = f1_gu.num_vars
    assert f3.num_vars == dim
    f3_f2u = scale(f3, f2_u)
    poly = ListOfProductsOfPolynomials.new(dim)
    poly.add_product([f1_gu, f3_f2u], field.ONE)
    return IPForMLSumcheck.prover_init(poly)


@dataclass
class GKRProof:  # data_structures.rs:9-19
    phase1_sumcheck_msgs: List[ProverMsg]
    phase2_sumcheck_msgs: List[ProverMsg]

    def extract_sum(self) -> np.ndarray:
        ev = self.phase1_sumcheck_msgs[0].evaluations
        return field.add(ev[0], ev[1])


@dataclass
class GKRRoundSumcheckSubClaim:  # data_structures.rs:22-31
    u: np.ndarray
    v: np.ndarray
    expected_evaluation: np.ndarray

    def verify_subclaim(self, f1: SparseMultilinearExtension, f2: DenseMultilinearExtension, f3: DenseMultilinearExtension, g) -> bool:
        """data_structures.rs:33-56"""
        dim = self.u.shape[0]
        assert self.v.shape[0] == dim and f1.num_vars == 3 * dim and f2.num_vars == dim and f3.num_vars == dim
        g = _np64(g).reshape(-1, 4)
        assert g.shape[0] == dim
        guv = np.concatenate([g, self.u, self.v])
        actual = field.to_int(f1.evaluate(guv)) * field.to_int(f2.evaluate(self.u)) % field.P * field.to_int(f3.evaluate(self.v)) % field.P
        return actual == field.to_int(self.expected_evaluation)


def _verify_phase(rng: Blake2b512Rng, msgs: Sequence[ProverMsg], dim: int, asserted_sum):
    """verifier_init{max_multiplicands: 2} + verify_round x dim + check_and_generate_subclaim (mod.rs:157-166)"""
    rs = []
    for i in range(dim):
        rng.feed(msgs[i])
        rs.append(rng.sample_fr())
    expected = field.to_int(asserted_sum)
    for i in range(dim):
        ev = msgs[i].evaluations
        if ev.shape[0] != 3:
            raise SumcheckError(5, "incorrect number of evaluations")
        try:  # field.to_int refuses limbs >= p: a proof element must be a canonical field element
            s01 = (field.to_int(ev[0]) + field.to_int(ev[1])) % field.P
        except ValueError:
            raise SumcheckError(5, "proof element is not a canonical field element")
        if s01 != expected:
            raise SumcheckError(8, "Prover message is not consistent with the claim.")
        expected = field.to_int(interpolate_uni_poly(ev, rs[i]))
    return np.stack(rs) if rs else np.zeros((0, 4), np.uint64), field.from_int(expected)


class GKRRoundSumcheck:
    @staticmethod
    def prove(rng: Blake2b512Rng, f1: SparseMultilinearExtension, f2: DenseMultilinearExtension, f3: DenseMultilinearExtension,
              g) -> GKRProof:
        """mod.rs:93-139: everything from the sparse fold to the last round polynomial runs inside sc_gkr_prove"""
        assert f1.num_vars == 3 * f2.num_vars and f1.num_vars == 3 * f3.num_vars
        dim = f2.num_vars
        g = _np64(g).reshape(-1, 4)
        assert g.shape[0] == dim
        proof = np.empty((2, max(dim, 1), 3, 4), dtype=np.uint64)
        dev = f1.on_device and f2.on_device and f3.on_device
        assert dev or not (f1.on_device or f2.on_device or f3.on_device), "mixing host and device inputs is not supported"
        if dev:
            import torch
            torch.cuda.current_stream(f2.evaluations.device).synchronize()  # the library works on its own stream
        i_ptr, v_ptr = f1._ptrs()
        check(lib().sc_gkr_prove(rng._h, i_ptr, v_ptr, f1.nnz, dim, _dense_ptr(f2), _dense_ptr(f3), _ptr(g),
                                 SC_TABLES_ON_DEVICE if dev else 0, _ptr(proof), None))
        return GKRProof([ProverMsg(proof[0, i].copy()) for i in range(dim)], [ProverMsg(proof[1, i].copy()) for i in range(dim)])

    @staticmethod
    def verify(rng: Blake2b512Rng, f2_num_vars: int, proof: GKRProof, claimed_sum) -> GKRRoundSumcheckSubClaim:
        """mod.rs:147-192"""
        dim = f2_num_vars
        u, exp1 = _verify_phase(rng, proof.phase1_sumcheck_msgs, dim, claimed_sum)
        v, exp2 = _verify_phase(rng, proof.phase2_sumcheck_msgs, dim, exp1)
        return GKRRoundSumcheckSubClaim(u, v, exp2)
