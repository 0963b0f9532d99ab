"""GKR initialisation with f1's non-zeros spread over several GPUs (reference
src/gkr_round_sumcheck/mod.rs:22-42, 57-63).

Every rank holds a disjoint subset of f1's (index, value) pairs -- any partition -- and all of f3.  a_hg[x] sums over ALL
non-zeros, so a rank's scatter is a partial sum; the ranks' dense tables are added with one table-sized integer all-reduce of
the widened limbs (2^dim x 8 uint64 lanes) and folded back mod p.  f1(g,.,.) stays distributed: each rank keeps the fold of its
own entries and folds / scatters it again in phase two, followed by the same all-reduce.  Exact field arithmetic => the same
canonical tables as the unsharded initialisation, whatever the partition.

Two drivers: the library does the all-reduce itself over an sc_comm (RCCL or a host transport) -- *_sharded below -- or leaves
the rank's lanes to the caller (`*_protocol`, over torch.distributed; the per-rank compute is pluggable there only so that the
collective logic can be tested on CPU with gloo)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import SC_TABLES_ON_DEVICE, check, lib
from .gkr_round_sumcheck import DenseMultilinearExtension, SparseMultilinearExtension, _dense_ptr, _np64, _ptr


def initialize_phase_one_sharded(comm, f1_local: SparseMultilinearExtension, f3: DenseMultilinearExtension, g):
    """-> (h_g complete on every rank, this rank's part of f1(g,.,.)); `comm`: sharded.NativeComm / P2PComm / HostComm (None = one rank).
    Host arrays, or -- all of f1_local and f3 -- torch tensors on the GPU: the library then reads them in place and leaves its outputs in
    device memory too (SC_TABLES_ON_DEVICE)."""
    dim = f3.num_vars
    assert f1_local.num_vars == 3 * dim
    on_dev = bool(f1_local.on_device and f3.on_device)
    assert on_dev or not (f1_local.on_device or f3.on_device), "f1 and f3 must both be host arrays or both be device tensors"
    g = _np64(g).reshape(-1, 4)
    nnz = f1_local.nnz
    n1 = C.c_uint64()
    i_ptr, v_ptr = f1_local._ptrs()
    h = comm._h if comm is not None else None
    if on_dev:
        import torch
        dev = f3.evaluations.device
        h_g = torch.empty((1 << dim, 4), dtype=torch.int64, device=dev)
        oi = torch.empty(max(nnz, 1), dtype=torch.int64, device=dev)
        ov = torch.empty((max(nnz, 1), 4), dtype=torch.int64, device=dev)
        check(lib().sc_gkr_phase_one_sharded(h, i_ptr, v_ptr, nnz, dim, _dense_ptr(f3), _ptr(g), SC_TABLES_ON_DEVICE, C.c_void_p(h_g.data_ptr()), None,
                                             C.c_void_p(oi.data_ptr()), C.c_void_p(ov.data_ptr()), C.byref(n1)))
        return DenseMultilinearExtension(dim, h_g), SparseMultilinearExtension(2 * dim, oi[: n1.value].contiguous(), ov[: n1.value].contiguous())
    h_g = np.empty((1 << dim, 4), dtype=np.uint64)
    oi = np.empty(max(nnz, 1), dtype=np.uint64)
    ov = np.empty((max(nnz, 1), 4), dtype=np.uint64)
    check(lib().sc_gkr_phase_one_sharded(h, i_ptr, v_ptr, nnz, dim, _dense_ptr(f3), _ptr(g), 0, _ptr(h_g), None, _ptr(oi), _ptr(ov), C.byref(n1)))
    return DenseMultilinearExtension(dim, h_g), SparseMultilinearExtension(2 * dim, oi[: n1.value].copy(), ov[: n1.value].copy())


def initialize_phase_two_sharded(comm, f1_g_local: SparseMultilinearExtension, u) -> DenseMultilinearExtension:
    u = _np64(u).reshape(-1, 4)
    dim = u.shape[0]
    assert f1_g_local.num_vars == 2 * dim
    i_ptr, v_ptr = f1_g_local._ptrs()
    h = comm._h if comm is not None else None
    if f1_g_local.on_device:
        import torch
        out = torch.empty((1 << dim, 4), dtype=torch.int64, device=f1_g_local.values.device)
        check(lib().sc_gkr_phase_two_sharded(h, i_ptr, v_ptr, f1_g_local.nnz, dim, _ptr(u), SC_TABLES_ON_DEVICE, C.c_void_p(out.data_ptr()), None))
        return DenseMultilinearExtension(dim, out)
    out = np.empty((1 << dim, 4), dtype=np.uint64)
    check(lib().sc_gkr_phase_two_sharded(h, i_ptr, v_ptr, f1_g_local.nnz, dim, _ptr(u), 0, _ptr(out), None))
    return DenseMultilinearExtension(dim, out)


def prove_sharded(comm, rng, f1_local: SparseMultilinearExtension, f2: DenseMultilinearExtension, f3: DenseMultilinearExtension, g):
    """GKRRoundSumcheck::prove over the ranks of `comm`, initialisations AND both sumcheck phases sharded (sc_gkr_prove_sharded): every rank
    passes its own subset of f1's non-zeros and all of f2, f3, g (host arrays, or all of them device tensors); every rank gets the same
    -> (phase-one messages (dim, 3, 4), phase-two messages (dim, 3, 4), u (dim, 4), v (dim, 4))."""
    dim = f3.num_vars
    assert f1_local.num_vars == 3 * dim and f2.num_vars == dim
    on_dev = bool(f1_local.on_device and f2.on_device and f3.on_device)
    assert on_dev or not (f1_local.on_device or f2.on_device or f3.on_device), "host arrays, or device tensors throughout"
    g = _np64(g).reshape(-1, 4)
    proof = np.empty((2, dim, 3, 4), dtype=np.uint64)
    uv = np.empty((2, dim, 4), dtype=np.uint64)
    i_ptr, v_ptr = f1_local._ptrs()
    check(lib().sc_gkr_prove_sharded(comm._h, rng._h, i_ptr, v_ptr, f1_local.nnz, dim, _dense_ptr(f2), _dense_ptr(f3), _ptr(g),
                                     SC_TABLES_ON_DEVICE if on_dev else 0, _ptr(proof), _ptr(uv)))
    return proof[0], proof[1], uv[0], uv[1]


class HipGkrEngine:
    """per-rank compute of the caller-driven protocol: the rank's contribution as lanes (sc_gkr_phase_*_sharded, lanes mode)"""

    def phase_one_partial(self, idx, vals, dim, f3, g):
        nnz = int(idx.shape[0])
        lanes = np.empty((1 << dim, 8), dtype=np.uint64)
        oi = np.empty(max(nnz, 1), dtype=np.uint64)
        ov = np.empty((max(nnz, 1), 4), dtype=np.uint64)
        n1 = C.c_uint64()
        check(lib().sc_gkr_phase_one_sharded(None, _ptr(idx), _ptr(vals), nnz, dim, _ptr(f3), _ptr(g), 0, None, _ptr(lanes), _ptr(oi), _ptr(ov), C.byref(n1)))
        return lanes, oi[: n1.value].copy(), ov[: n1.value].copy()

    def phase_two_partial(self, idx, vals, dim, u):
        lanes = np.empty((1 << dim, 8), dtype=np.uint64)
        check(lib().sc_gkr_phase_two_sharded(None, _ptr(idx), _ptr(vals), int(idx.shape[0]), dim, _ptr(u), 0, None, _ptr(lanes)))
        return lanes

    def fold(self, lanes):
        out = np.empty((lanes.shape[0], 4), dtype=np.uint64)
        check(lib().sc_wide_reduce_table(_ptr(lanes), lanes.shape[0], _ptr(out), 0))
        return out


def _all_reduce_lanes(dist_comm, lanes: np.ndarray) -> np.ndarray:
    import torch
    t = torch.from_numpy(lanes.view(np.int64).copy())
    dist_comm.all_reduce_sum(t)
    return t.numpy().view(np.uint64)


def phase_one_protocol(engine, dist_comm, idx_local, vals_local, dim, f3, g):
    """caller-driven initialize_phase_one over torch.distributed (sharded.DistComm): -> (h_g, local f1_g idx, vals)"""
    idx_local = np.ascontiguousarray(idx_local, dtype=np.uint64)
    vals_local, f3, g = _np64(vals_local).reshape(-1, 4), _np64(f3).reshape(-1, 4), _np64(g).reshape(-1, 4)
    lanes, oi, ov = engine.phase_one_partial(idx_local, vals_local, dim, f3, g)
    return engine.fold(_all_reduce_lanes(dist_comm, lanes)), oi, ov


def phase_two_protocol(engine, dist_comm, f1g_idx_local, f1g_vals_local, dim, u):
    f1g_idx_local = np.ascontiguousarray(f1g_idx_local, dtype=np.uint64)
    lanes = engine.phase_two_partial(f1g_idx_local, _np64(f1g_vals_local).reshape(-1, 4), dim, _np64(u).reshape(-1, 4))
    return engine.fold(_all_reduce_lanes(dist_comm, lanes))
