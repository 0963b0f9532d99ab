"""world_size-2 `gloo` test of the multi-GPU sharded protocol on CPU.

The collective logic under test is the product's (sumcheck_amd/sharded.py: integer all-reduce of widened limbs,
sc_wide_reduce, bind_final + all-gather tail, transcript replicated per rank).  The per-shard compute engine is
swapped for the CPU oracle here -- in tests only -- because there is no GPU in the build container."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cref
from tests import helpers as H


class OracleShardEngine:
    """test double with the HipShardEngine interface, backed by oracle/liboracle.so"""

    def __init__(self, nv_local, shapes, coeffs, tables):
        tabs = [np.ascontiguousarray(t.numpy().view(np.uint64) if isinstance(t, torch.Tensor) else t, dtype=np.uint64) for t in tables]
        self.desc = cref.PolyDesc(nv_local, [(coeffs[k], list(s)) for k, s in enumerate(shapes)], tabs)
        self.p = cref.Prover(self.desc)
        self.last_r = None

    def round_partial(self, r):
        ev = self.p.prove_round(r)  # (D,4) u64 -> (D,8) zero-extended 32-bit limbs
        lanes = ev.view(np.uint32).reshape(ev.shape[0], 8).astype(np.int64)
        return torch.from_numpy(lanes)

    def bind_final(self, r):
        _, tabs, _ = self.p.state()  # (U, 2, 4)
        out = np.stack([cref.fix_variables(tabs[u], r.reshape(1, 4))[0] for u in range(tabs.shape[0])])
        return torch.from_numpy(out.view(np.int64))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nv, shapes, nt, L, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sumcheck_amd import sharded
    G = world * L
    n_loc = (1 << nv) // G
    tabs = [cref.synth_table(91, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(91, 1000, len(shapes))
    engines = [OracleShardEngine(nv - (G.bit_length() - 1), shapes, coefs,
                                 [t[(rank * L + l) * n_loc:(rank * L + l + 1) * n_loc] for t in tabs]) for l in range(L)]
    tail = lambda nvt, tables: OracleShardEngine(nvt, shapes, coefs, [tables[u] for u in range(tables.shape[0])])
    proof, rand = sharded.prove_sharded(engines, sharded.DistComm(), nv, max(len(s) for s in shapes), tail)
    q.put((rank, proof, rand))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nv,shapes,nt,L", [
    (8, [[0, 1, 2]], 3, 1),                 # 2 shards, BASELINE C4 product shape
    (7, [[0, 1, 2], [1, 3], [2]], 4, 2),    # 4 shards (2 ranks x 2 logical), shared tables
    (2, [[0, 1]], 2, 1),                    # smallest legal: every shard ends with one pair
])
def test_gloo_world2_matches_unsharded_oracle(nv, shapes, nt, L):
    tabs = [cref.synth_table(91, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(91, 1000, len(shapes))
    d = cref.PolyDesc(nv, [(coefs[k], list(s)) for k, s in enumerate(shapes)], tabs)
    want, wrand = cref.ml_prove(d)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nv, shapes, nt, L, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, proof, rand in res:
        assert np.array_equal(proof, want), f"rank {rank}"
        assert np.array_equal(rand, wrand), f"rank {rank}"


def _transport_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sumcheck_amd import sharded
        comm = sharded.HostComm.over_torch_distributed()
        comm.selftest()
        comm.selftest()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_library_host_transport_over_gloo_world2():
    """the library's host-transport communicator (sc_comm_init_host), the path sc_ml_prove_sharded takes without RCCL, driven
    through the C ABI by two gloo ranks: its all-reduce and all-gather callbacks deliver what the library expects"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transport_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: "ok", 1: "ok"}


def test_library_host_transport_between_threads():
    """the same with one thread per rank inside one process (the process model a Rust host would use)"""
    import threading
    from sumcheck_amd import sharded
    world = 4
    ex = sharded.ThreadExchange(world)
    errs = []

    def run(rank):
        try:
            c = ex.comm(rank)
            c.selftest()
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errs, errs


def test_library_host_transport_between_threads_with_delays_and_reordering():
    """gloo-free lock-order shake-out of the library's host-transport path: eight thread ranks run the collective self-test 40 times
    over a transport stub that sleeps a random time before and after every exchange, so that the ranks enter the library's
    collectives (which let go of the per-device gate around the callbacks: GateYield) in a different order every time, while other
    threads of the process hammer the same gate through host-only entry points.  Nothing may deadlock, every sum must be right."""
    import random
    import threading
    import time
    from sumcheck_amd import sharded
    import sumcheck_amd as sc
    world, rounds = 8, 40
    ex = sharded.ThreadExchange(world)
    rnd = random.Random(11)
    errs, done = [], threading.Event()

    def comm_for(rank):
        def allreduce(a):
            time.sleep(rnd.random() * 1e-3)
            parts = ex._exchange(rank, a)
            time.sleep(rnd.random() * 5e-4)
            tot = np.zeros_like(a)
            for x in parts:
                tot += x
            return tot

        def allgather(b):
            time.sleep(rnd.random() * 1e-3)
            got = np.concatenate(ex._exchange(rank, b))
            time.sleep(rnd.random() * 5e-4)
            return got
        return sharded.HostComm(rank, world, allreduce, allgather)

    def run(rank):
        try:
            c = comm_for(rank)
            for _ in range(rounds):
                c.selftest()
            c.close()
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))

    def bystander():  # host-only library calls from other threads while the collectives run
        r = sc.Blake2b512Rng.setup()
        while not done.is_set():
            r.feed(b"x" * 200)
            r.sample_fr()
            sc.lib().sc_release_caches()

    by = [threading.Thread(target=bystander) for _ in range(2)]
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in by + ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    done.set()
    for t in by:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in ts), "a rank is stuck"
    assert not errs, errs


def test_sharded_entry_point_validates_before_touching_a_device():
    import ctypes as C
    import sumcheck_amd as sc
    from sumcheck_amd import _lib, sharded
    c = sharded.HostComm(0, 1, lambda a: a, lambda b: b)
    out = np.zeros((4, 4), np.uint64)
    rc = sc.lib().sc_ml_prove_sharded(None, c._h, None, 4, C.c_void_p(out.ctypes.data), C.c_void_p(out.ctypes.data))
    assert rc == _lib.SC_ERR_BAD_ARG
    h = C.c_void_p()
    assert sc.lib().sc_comm_init_host(2, 2, None, None, None, C.byref(h)) == _lib.SC_ERR_BAD_ARG  # rank out of range
    assert sc.lib().sc_comm_init_host(0, 2, None, None, None, C.byref(h)) == _lib.SC_ERR_BAD_ARG  # transports required for > 1 rank


class OracleGkrEngine:
    """test double of sharded_gkr.HipGkrEngine: the rank's partial tables from the CPU oracle, widened to lanes; the fold is the
    product's host routine (sc_wide_reduce)"""

    @staticmethod
    def _widen(tab):
        return np.ascontiguousarray(tab.view(np.uint32).reshape(tab.shape[0], 8).astype(np.uint64))

    def phase_one_partial(self, idx, vals, dim, f3, g):
        h, oi, ov = cref.gkr_phase_one(idx, vals, dim, f3, g)
        return self._widen(h), oi, ov

    def phase_two_partial(self, idx, vals, dim, u):
        return self._widen(cref.gkr_phase_two(idx, vals, dim, u))

    def fold(self, lanes):
        from sumcheck_amd import sharded
        return sharded.wide_reduce(lanes)


def _gkr_inputs(dim, seed):
    rng = np.random.default_rng(seed)
    n = 1 << dim
    idx = np.unique((rng.integers(0, 1 << dim, size=3 * n, dtype=np.uint64)) | (rng.integers(0, 1 << dim, size=3 * n, dtype=np.uint64) << np.uint64(dim))
                    | (rng.integers(0, 4, size=3 * n, dtype=np.uint64) << np.uint64(2 * dim)))[: 2 * n]  # many (x, y) collisions across z
    return idx, cref.synth_table(seed, 1, idx.shape[0]), cref.synth_table(seed, 3, n), cref.synth_table(seed, 4, dim), cref.synth_table(seed, 5, dim)


def _gkr_worker(rank, world, port, dim, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sumcheck_amd import sharded, sharded_gkr
        idx, vals, f3, g, u = _gkr_inputs(dim, 321)
        mine = slice(rank, None, world)  # a strided partition: the keys of f1(g,.,.) overlap across ranks
        eng, comm = OracleGkrEngine(), sharded.DistComm()
        h_g, gi, gv = sharded_gkr.phase_one_protocol(eng, comm, idx[mine], vals[mine], dim, f3, g)
        f1_gu = sharded_gkr.phase_two_protocol(eng, comm, gi, gv, dim, u)
        q.put((rank, h_g, f1_gu))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gloo_world2_sharded_gkr_initialisation_matches_oracle():
    """f4: initialize_phase_one / _two with f1's non-zeros split over two ranks (table-sized widened all-reduce) equal the
    unsharded oracle's tables"""
    dim = 6
    idx, vals, f3, g, u = _gkr_inputs(dim, 321)
    wh, wi, wv = cref.gkr_phase_one(idx, vals, dim, f3, g)
    wgu = cref.gkr_phase_two(wi, wv, dim, u)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gkr_worker, args=(r, 2, port, dim, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, h_g, f1_gu in res:
        assert np.array_equal(h_g, wh), f"rank {rank}: h_g"
        assert np.array_equal(f1_gu, wgu), f"rank {rank}: f1(g,u,.)"
