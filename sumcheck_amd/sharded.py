"""Multi-GPU sharded prover: one process per GPU, tables partitioned by the HIGH index bits.

Because the protocol binds variables LSB first (reference src/ml_sumcheck/protocol/prover.rs:119-120 pairs
entries 2b, 2b+1), a contiguous block of 2^(nv - log2 G) entries of every table is itself a table over the low
nv - log2 G variables, and stays contiguous under every halving.  Per round each shard computes its partial
round polynomial with the ordinary kernels; the only exchange is an integer all-reduce (RCCL over xGMI via
torch.distributed; `gloo` on CPU) of the (deg+1) x 8 zero-extended 32-bit limbs -- 64 bytes per evaluation --
which every rank folds back into the field identically (sc_wide_reduce), feeds to its own copy of the
transcript and so derives the same challenge with no further communication.  When a shard is down to one
element per table (after nv - log2 G rounds) the G elements are all-gathered and the last log2 G rounds run
on a G-entry table on every rank.

The per-shard compute engine is pluggable only so that the collective logic can be unit-tested on CPU with
`gloo`; the product engine is HipShardEngine (libsumcheck_hip.so) and there is no default fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import field
from ._lib import SC_TABLES_BORROW, SC_TABLES_ON_DEVICE, PolyDesc, check, lib
from .ml_sumcheck import Blake2b512Rng, PolynomialInfo, ProverMsg


def _log2(x: int) -> int:
    assert x > 0 and x & (x - 1) == 0, "shard count must be a power of two"
    return x.bit_length() - 1


def wide_reduce(wide: np.ndarray) -> np.ndarray:
    """(n,8) summed uint64 lanes -> (n,4) canonical Montgomery limbs (host; sc_wide_reduce)"""
    wide = np.ascontiguousarray(wide, dtype=np.uint64).reshape(-1, 8)
    out = np.empty((wide.shape[0], 4), dtype=np.uint64)
    check(lib().sc_wide_reduce(C.c_void_p(wide.ctypes.data), wide.shape[0], C.c_void_p(out.ctypes.data)))
    return out


class HipShardEngine:
    """One shard on one GPU: an sc_prover over the shard's local tables, driven through
    sc_prove_round_partial / sc_prover_bind_final on torch's current stream."""

    def __init__(self, nv_local: int, shapes: Sequence[Sequence[int]], coeffs: np.ndarray, tables, device, borrow: bool = True,
                 streamed_chunk_log2: Optional[int] = None):
        """streamed_chunk_log2 (0 = the library's default): the shard's tables are HOST arrays that stay in host memory and are pulled
        through HBM chunk by chunk in rounds 1 and 2 (sc_prover_init_streamed): out-of-core x multi-GPU."""
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.nv = nv_local
        self.U = len(tables)
        self.D = max(len(s) for s in shapes) + 1
        self._tables = []
        self._streamed = streamed_chunk_log2 is not None
        for t in tables:
            if self._streamed:
                t = t.numpy().view(np.uint64) if isinstance(t, torch.Tensor) else t
                self._tables.append(np.ascontiguousarray(t, dtype=np.uint64))
                continue
            if not isinstance(t, torch.Tensor):
                t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.uint64).view(np.int64))
            self._tables.append(t.to(self.device).contiguous())
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        offs, idx = [0], []
        for s in shapes:
            idx.extend(int(i) for i in s)
            offs.append(len(idx))
        offsets = np.asarray(offs, dtype=np.uint32)
        indices = np.asarray(idx, dtype=np.uint32)
        tabs = (C.c_void_p * self.U)(*[(t.ctypes.data if self._streamed else t.data_ptr()) for t in self._tables])
        d = PolyDesc()
        d.num_vars, d.max_multiplicands, d.n_products = nv_local, self.D - 1, len(shapes)
        d.coeffs = coeffs.ctypes.data_as(C.POINTER(C.c_uint64))
        d.prod_offsets = offsets.ctypes.data_as(C.POINTER(C.c_uint32))
        d.prod_indices = indices.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_tables = self.U
        d.tables = C.cast(tabs, C.POINTER(C.c_void_p))
        d.flags = 0 if self._streamed else (SC_TABLES_ON_DEVICE | (SC_TABLES_BORROW if borrow else 0))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().sc_set_device(self.device.index or 0))
            if self._streamed:
                check(lib().sc_prover_init_streamed(C.byref(d), int(streamed_chunk_log2), C.byref(self._h)))
            else:
                check(lib().sc_prover_init(C.byref(d), C.byref(self._h)))
            check(lib().sc_prover_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), 0))

    def round_partial(self, r: Optional[np.ndarray]):
        """-> (D,8) int64 tensor on the GPU: zero-extended 32-bit limbs of this shard's partial evaluations"""
        out = self.torch.empty((self.D, 8), dtype=self.torch.int64, device=self.device)
        rp = C.c_void_p(np.ascontiguousarray(r, dtype=np.uint64).ctypes.data) if r is not None else None
        check(lib().sc_prove_round_partial(self._h, rp, C.c_void_p(out.data_ptr())))
        return out

    def round_full(self, r: Optional[np.ndarray]) -> np.ndarray:
        """-> (D,4) uint64: the round's canonical message (sc_prove_round); for provers that hold whole tables (the tail)"""
        out = np.empty((self.D, 4), dtype=np.uint64)
        rp = C.c_void_p(np.ascontiguousarray(r, dtype=np.uint64).ctypes.data) if r is not None else None
        check(lib().sc_prove_round(self._h, rp, C.c_void_p(out.ctypes.data)))
        return out

    def prove_rounds(self, rng: Blake2b512Rng, n_rounds: int):
        """-> (messages (n_rounds, D, 4), challenges (n_rounds, 4)): n_rounds x (prove_round, feed, sample) inside the library,
        continuing `rng` (sc_ml_prove_rounds); for provers that hold whole tables (the tail)"""
        msgs = np.empty((n_rounds, self.D, 4), dtype=np.uint64)
        ch = np.empty((n_rounds, 4), dtype=np.uint64)
        torch = self.torch
        # on the handle's own non-blocking stream (see prove_sharded_native): its pipelined rounds park a polling kernel there
        check(lib().sc_prover_set_stream(self._h, None, 1))  # synchronises the torch stream the tables were copied on
        try:
            check(lib().sc_ml_prove_rounds(self._h, rng._h, n_rounds, C.c_void_p(msgs.ctypes.data), C.c_void_p(ch.ctypes.data)))
        finally:
            check(lib().sc_prover_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), 0))
        return msgs, ch

    def bind_final(self, r: np.ndarray):
        """-> (U,4) int64 tensor: the single remaining element of every local table"""
        out = self.torch.empty((self.U, 4), dtype=self.torch.int64, device=self.device)
        check(lib().sc_prover_bind_final(self._h, C.c_void_p(np.ascontiguousarray(r, dtype=np.uint64).ctypes.data),
                                         C.c_void_p(out.data_ptr())))
        return out

    def reset(self):
        """rewind to round 0 over the same resident shard (borrowing handles only)"""
        check(lib().sc_prover_reset(self._h, None, 0))

    def reload(self, tables):
        """rewind to round 0 over NEW tables of the same shape (device tensors): a copying handle copies them in again, a
        borrowing one re-points.  No allocation -- this is what keeps the per-proof tail prover cheap."""
        torch = self.torch
        self._tables = [t.to(self.device).contiguous() for t in tables]
        assert len(self._tables) == self.U
        tabs = (C.c_void_p * self.U)(*[t.data_ptr() for t in self._tables])
        with torch.cuda.device(self.device):
            check(lib().sc_prover_reset(self._h, C.cast(tabs, C.POINTER(C.c_void_p)), SC_TABLES_ON_DEVICE))

    def close(self):
        if self._h:
            self.torch.cuda.synchronize(self.device)
            lib().sc_prover_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TailEngines:
    """tail_factory for prove_sharded / prove_sharded_native that builds the log2(G)-variable tail prover once and reloads it
    on later proofs (prover_init costs a dozen allocations; a reload is U small device copies)."""

    def __init__(self, shapes, coeffs, device):
        self.shapes, self.coeffs, self.device = shapes, coeffs, device
        self._engines = {}

    def __call__(self, nvt, tabs):
        tl = [tabs[u] for u in range(tabs.shape[0])]
        e = self._engines.get(nvt)
        if e is None:
            e = HipShardEngine(nvt, self.shapes, self.coeffs, tl, self.device, borrow=False)
            self._engines[nvt] = e
        else:
            e.reload(tl)
        return e


class DistComm:
    """torch.distributed world ("nccl" = RCCL on ROCm, "gloo" on CPU)"""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0

    def _host_backend(self) -> bool:
        """gloo moves host tensors only: device tensors take a round trip through the host (tests: two processes on one GPU)"""
        return self.world > 1 and self.dist.get_backend() == "gloo"

    def all_reduce_sum(self, t):
        if self.world > 1:
            if self._host_backend() and t.is_cuda:
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def all_gather(self, t):
        import torch
        if self.world == 1:
            return t.unsqueeze(0)
        if self._host_backend() and t.is_cuda:
            h = t.cpu()
            out = [torch.empty_like(h) for _ in range(self.world)]
            self.dist.all_gather(out, h)
            return torch.stack(out).to(t.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return torch.stack(out)


def prove_sharded(engines: Sequence, comm: DistComm, nv_total: int, max_multiplicands: int, tail_factory,
                  fs_rng: Optional[Blake2b512Rng] = None):
    """MLSumcheck::prove_as_subprotocol (reference src/ml_sumcheck/mod.rs:50-70) over G = len(engines) * comm.world
    shards.  `engines` are this process's shards in increasing global order (normally exactly one per GPU).
    tail_factory(nv_tail, tables[(U, G, 4) int64 tensor]) builds the engine for the last log2 G rounds.
    Returns (proof (nv, D, 4) uint64, randomness (nv, 4) uint64); identical on every rank."""
    import torch
    L = len(engines)
    G = L * comm.world
    k = _log2(G)
    nv_local = nv_total - k
    assert nv_local >= 1, "each shard needs at least two entries per table"
    rng = fs_rng or Blake2b512Rng.setup()
    rng.feed(PolynomialInfo(max_multiplicands, nv_total))  # mod.rs:54
    D = max_multiplicands + 1
    proof = np.empty((nv_total, D, 4), dtype=np.uint64)
    rand = np.empty((nv_total, 4), dtype=np.uint64)
    r = None
    for i in range(nv_local):
        wide = engines[0].round_partial(r)
        for e in engines[1:]:
            wide = wide + e.round_partial(r)
        wide = comm.all_reduce_sum(wide)
        evals = wide_reduce(wide.cpu().numpy().view(np.uint64))
        proof[i] = evals
        rng.feed(ProverMsg(evals))  # mod.rs:61
        r = rng.sample_fr()         # mod.rs:63
        rand[i] = r
    if k > 0:
        local = torch.stack([e.bind_final(r) for e in engines])        # (L, U, 4)
        allsh = comm.all_gather(local)                                  # (world, L, U, 4)
        U = local.shape[1]
        tables = allsh.reshape(G, U, 4).permute(1, 0, 2).contiguous()   # (U, G, 4): entry g of table u came from shard g
        tail = tail_factory(k, tables)
        _run_tail(tail, rng, k, proof[nv_local:], rand[nv_local:])
    return proof, rand


def _run_tail(tail, rng, k, proof_out, rand_out):
    """the last k rounds, on the gathered tables every rank holds in full: one library call for a HIP engine, the per-round
    loop otherwise (test doubles)"""
    if hasattr(tail, "prove_rounds"):
        msgs, ch = tail.prove_rounds(rng, k)
        proof_out[:k] = msgs
        rand_out[:k] = ch
        return
    rt = None
    for j in range(k):
        evals = _tail_round(tail, rt)
        proof_out[j] = evals
        rng.feed(ProverMsg(evals))
        rt = rng.sample_fr()
        rand_out[j] = rt


def _tail_round(tail, rt):
    """one round of the tail prover: every rank holds the complete G-entry tables, so the round needs no exchange; a HIP engine
    returns the canonical message straight from sc_prove_round (polled host-mapped result), other engines go through the
    widened partial form"""
    if hasattr(tail, "round_full"):
        return tail.round_full(rt)
    p = tail.round_partial(rt)
    return wide_reduce((p.cpu().numpy() if hasattr(p, "cpu") else np.asarray(p)).view(np.uint64))


class NativeComm:
    """An RCCL communicator owned by libsumcheck_hip.so (sc_comm): the per-round all-reduce then runs inside the library on
    the prover's stream, with no Python between rounds.  The 128-byte unique id travels over torch.distributed."""

    def __init__(self, device):
        import torch
        import torch.distributed as dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        idb = (C.c_uint8 * 128)()
        if self.rank == 0:
            check(lib().sc_comm_unique_id(C.cast(idb, C.c_void_p)))
        t = torch.tensor(list(bytes(idb)), dtype=torch.uint8, device=device)
        if self.world > 1:
            dist.broadcast(t, src=0)
        idb = (C.c_uint8 * 128)(*t.cpu().tolist())
        self._h = C.c_void_p()
        check(lib().sc_set_device(torch.device(device).index or 0))
        check(lib().sc_comm_init(C.cast(idb, C.c_void_p), self.rank, self.world, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().sc_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class P2PComm:
    """A peer-to-peer communicator (sc_comm_init_p2p): the ranks are THREADS of this process with one GPU each (or, in functional
    tests, sharing one); a round's all-reduce is one kernel per rank writing into its peers' inboxes.  Every rank's thread calls this
    constructor with the same group id (the call blocks until all have)."""

    def __init__(self, group_id: int, rank: int, world: int, device=None):
        self.rank, self.world = rank, world
        if device is not None:
            import torch
            check(lib().sc_set_device(torch.device(device).index or 0))
        self._h = C.c_void_p()
        check(lib().sc_comm_init_p2p(group_id, rank, world, C.byref(self._h)))

    def selftest(self):
        check(lib().sc_comm_selftest(self._h))

    def close(self):
        if self._h:
            lib().sc_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostComm:
    """A HOST-transport communicator (sc_comm_init_host): the library hands the (deg+1) x 8 uint64 lanes of a round, and the
    tail's U x 32 bytes, to two Python callables that exchange host buffers.  Two transports are provided: torch.distributed
    (any backend that moves host tensors: gloo) and an in-process one for one-thread-per-GPU use (ThreadExchange)."""

    def __init__(self, rank: int, world: int, allreduce, allgather):
        self.rank, self.world = rank, world
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t)
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

        def _ar(_ctx, inout, count):
            try:
                a = np.ctypeslib.as_array(inout, shape=(count,))
                a[:] = allreduce(a.copy())
                return 0
            except Exception:  # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1

        def _ag(_ctx, send, recv, nbytes):
            try:
                sbuf = np.frombuffer(C.string_at(send, nbytes), dtype=np.uint8)
                out = np.ascontiguousarray(allgather(sbuf), dtype=np.uint8).reshape(-1)
                assert out.shape[0] == nbytes * world
                C.memmove(recv, out.ctypes.data, nbytes * world)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        self._ar, self._ag = AR(_ar), AG(_ag)  # keep the trampolines alive as long as the communicator
        self._h = C.c_void_p()
        check(lib().sc_comm_init_host(rank, world, C.cast(self._ar, C.c_void_p), C.cast(self._ag, C.c_void_p), None, C.byref(self._h)))

    def selftest(self):
        """collective: exercises both transport functions with known patterns (sc_comm_selftest)"""
        check(lib().sc_comm_selftest(self._h))

    @classmethod
    def over_torch_distributed(cls):
        import torch
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0

        def allreduce(a):
            if world == 1:
                return a
            t = torch.from_numpy(a.view(np.int64).copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy().view(np.uint64)

        def allgather(b):
            if world == 1:
                return b
            t = torch.from_numpy(b.copy())
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)
            return torch.cat(out).numpy()

        return cls(rank, world, allreduce, allgather)

    def close(self):
        if self._h:
            lib().sc_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ThreadExchange:
    """In-process transport for one host thread per shard (the process model a Rust host would use: sc_set_device per thread,
    one sc_prover per thread): a barrier-synchronised slot array.  comm(rank) -> HostComm for that thread."""

    def __init__(self, world: int):
        import threading
        self.world = world
        self._slots = [None] * world
        self._bar = threading.Barrier(world)

    def _exchange(self, rank, value):
        self._slots[rank] = value
        self._bar.wait(timeout=120)
        got = list(self._slots)
        self._bar.wait(timeout=120)  # nobody overwrites a slot before everybody has read it
        return got

    def comm(self, rank: int) -> HostComm:
        def allreduce(a):
            parts = self._exchange(rank, a)
            tot = np.zeros_like(a)
            for x in parts:
                tot += x
            return tot

        def allgather(b):
            return np.concatenate(self._exchange(rank, b))

        return HostComm(rank, self.world, allreduce, allgather)


def prove_sharded_library(engine: HipShardEngine, comm, nv_total: int, fs_rng: Optional[Blake2b512Rng] = None):
    """The whole sharded proof as ONE library call (sc_ml_prove_sharded): local rounds with the per-round all-reduce, bind_final,
    all-gather and the log2 G tail rounds all run inside libsumcheck_hip.so.  `comm` is a NativeComm (RCCL) or a HostComm.
    -> (proof (nv_total, D, 4), randomness (nv_total, 4)), identical on every rank."""
    import torch
    D = engine.D
    proof = np.empty((nv_total, D, 4), dtype=np.uint64)
    rand = np.empty((nv_total, 4), dtype=np.uint64)
    # The library loop runs on the handle's own non-blocking stream: its pipelined late rounds park a polling wait kernel on the
    # stream, which must not be torch's (possibly legacy-default, implicitly synchronising) stream.  Both switches synchronise.
    torch.cuda.current_stream(engine.device).synchronize()
    check(lib().sc_prover_set_stream(engine._h, None, 1))
    try:
        check(lib().sc_ml_prove_sharded(engine._h, comm._h, fs_rng._h if fs_rng is not None else None, nv_total, C.c_void_p(proof.ctypes.data),
                                        C.c_void_p(rand.ctypes.data)))
    finally:
        check(lib().sc_prover_set_stream(engine._h, C.c_void_p(torch.cuda.current_stream(engine.device).cuda_stream), 0))
    return proof, rand


def prove_sharded_native(engine: HipShardEngine, ncomm, comm: DistComm, nv_total: int, max_multiplicands: int, tail_factory=None,
                         fs_rng: Optional[Blake2b512Rng] = None):
    """kept name of the round-1 driver: now a thin caller of sc_ml_prove_sharded (the tail no longer runs in Python)"""
    assert max_multiplicands + 1 == engine.D
    return prove_sharded_library(engine, ncomm, nv_total, fs_rng)


def prove_logical_shards(nv: int, shapes, tables: Sequence[np.ndarray], coeffs: np.ndarray, G: int, device):
    """Single-process, single-GPU run of the sharded protocol with G logical shards (tests / debugging)."""
    k = _log2(G)
    n_loc = 1 << (nv - k)
    engines = [HipShardEngine(nv - k, shapes, coeffs, [t[g * n_loc:(g + 1) * n_loc] for t in tables], device, borrow=True)
               for g in range(G)]
    return prove_sharded(engines, DistComm(), nv, max(len(s) for s in shapes), TailEngines(shapes, coeffs, device))
