"""GPU tests of the multi-GPU proof behind the C ABI (sc_ml_prove_sharded): local rounds, per-round all-reduce, bind + all-gather
and the log2 G tail rounds all inside libsumcheck_hip.so.  On a one-GPU box the G shards share the GPU and meet through the
library's HOST transport (threads of one process, or processes over gloo); with N visible GPUs the same entry point runs over
RCCL, one rank per GPU, as processes and as threads.  Every proof is compared bit for bit with the unsharded oracle proof."""
import ctypes as C
import os
import socket
import threading

import numpy as np
import pytest

import sumcheck_amd as sc
from oracle import cref
from sumcheck_amd import _lib, field, sharded
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _n_gpus():
    return sc.lib().sc_device_count()


def _oracle(nv, shapes, nt, seed):
    tabs = [cref.synth_table(seed, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(seed, 1000, len(shapes))
    want, wrand = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=cref.max_threads())
    return tabs, coefs, want, wrand


def _thread_rank(rank, G, nv, shapes, tabs, coefs, make_comm, device, out, n_proofs=2):
    """one host thread = one rank: its own device selection, prover handle and communicator"""
    try:
        import torch
        _lib.check(sc.lib().sc_set_device(device))
        n_loc = (1 << nv) // G
        with torch.cuda.device(device):
            eng = sharded.HipShardEngine(nv - (G.bit_length() - 1), shapes, coefs, [t[rank * n_loc:(rank + 1) * n_loc] for t in tabs],
                                         f"cuda:{device}", borrow=True)
            comm = make_comm(rank)
            res = []
            for _ in range(n_proofs):  # the second proof reuses the handle, the tail prover and every buffer
                eng.reset()
                res.append(sharded.prove_sharded_library(eng, comm, nv))
            comm.close()
            eng.close()
        out[rank] = res
    except Exception as e:  # noqa: BLE001
        import traceback
        out[rank] = RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")


@pytest.mark.parametrize("G,nv,nt,shapes", [
    (2, 13, 4, [[0, 1, 2], [3, 3], [1]]),
    (4, 19, 10, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]),   # config-3 shape; 2^17 entries per shard: no big rounds
    (2, 20, 3, [[0, 1, 2]]),                               # config-4 shape; 2^19 per shard: two big rounds (merged kernel, F29) per shard
    (8, 6, 2, [[0, 1], [1]]),                              # tail as long as the local part
])
def test_sharded_proof_threads_on_one_gpu(G, nv, nt, shapes):
    """single process, one thread per rank (sc_set_device per thread -- how a Rust host drives it), all on GPU 0, meeting through
    the in-process host transport"""
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 4100 + nv)
    ex = sharded.ThreadExchange(G)
    out = [None] * G
    ts = [threading.Thread(target=_thread_rank, args=(r, G, nv, shapes, tabs, coefs, ex.comm, 0, out)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(G):
        assert not isinstance(out[r], Exception), out[r]
        for proof, rand in out[r]:
            assert np.array_equal(proof, want), f"rank {r}"
            assert np.array_equal(rand, wrand), f"rank {r}"


_P2P_GROUP = [7000]


@pytest.mark.parametrize("G,nv,nt,shapes", [
    (2, 13, 4, [[0, 1, 2], [3, 3], [1]]),
    (4, 19, 10, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]),
    (2, 20, 3, [[0, 1, 2]]),
    (8, 12, 2, [[0, 1], [1]]),
])
def test_sharded_proof_peer_to_peer_exchange_threads_on_one_gpu(G, nv, nt, shapes):
    """sc_comm_init_p2p: the per-round all-reduce as one kernel per rank pushing self-validating words into its peers' inboxes (no RCCL,
    no host transport), the tail's gather as peer copies.  Functionally on ONE GPU: the G thread ranks share it, so the exchange kernel
    gives up quickly when a peer's kernels have not run yet and the host launches it again.  Includes the group's self-test (peer
    all-gather + table-sized peer all-reduce).  Every rank's proof equals the unsharded oracle proof."""
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 4300 + nv)
    _P2P_GROUP[0] += 1
    gid = _P2P_GROUP[0]
    out = [None] * G

    def mk(rank):
        c = sharded.P2PComm(gid, rank, G, "cuda:0")
        c.selftest()
        return c

    ts = [threading.Thread(target=_thread_rank, args=(r, G, nv, shapes, tabs, coefs, mk, 0, out)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(G):
        assert not isinstance(out[r], Exception) and out[r] is not None, out[r]
        for proof, rand in out[r]:
            assert np.array_equal(proof, want), f"rank {r}"
            assert np.array_equal(rand, wrand), f"rank {r}"


def test_peer_to_peer_group_misuse_is_refused():
    """a rank that joins a group twice, ranks that disagree on its size, a size beyond the inbox layout, a rank outside the group"""
    _P2P_GROUP[0] += 1
    gid = _P2P_GROUP[0]
    _lib.check(sc.lib().sc_set_device(0))
    a = sharded.P2PComm(gid, 0, 1, "cuda:0")  # a one-rank group forms at once
    a.selftest()
    h = C.c_void_p()
    assert sc.lib().sc_comm_init_p2p(gid, 0, 1, C.byref(h)) == _lib.SC_ERR_BAD_ARG       # joined twice
    assert sc.lib().sc_comm_init_p2p(gid, 1, 2, C.byref(h)) == _lib.SC_ERR_BAD_ARG       # disagrees on the size
    assert sc.lib().sc_comm_init_p2p(gid + 100000, 0, 64, C.byref(h)) == _lib.SC_ERR_BAD_ARG
    assert sc.lib().sc_comm_init_p2p(gid + 100000, 3, 2, C.byref(h)) == _lib.SC_ERR_BAD_ARG
    a.close()
    b = sharded.P2PComm(gid, 0, 1, "cuda:0")  # the id is free again once the last rank has left
    b.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_proof_peer_to_peer_one_thread_per_gpu(world):
    """the same on distinct GPUs (xGMI peer writes, pipelined late rounds): skipped below `world` visible GPUs"""
    if _n_gpus() < world:
        pytest.skip(f"needs {world} visible GPUs")
    nv, shapes, nt = 21, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 97)
    _P2P_GROUP[0] += 1
    gid = _P2P_GROUP[0]
    out = [None] * world
    ts = [threading.Thread(target=lambda r=r: _thread_rank(r, world, nv, shapes, tabs, coefs, lambda rank: sharded.P2PComm(gid, rank, world, f"cuda:{rank}"), r, out))
          for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        for proof, rand in out[r]:
            assert np.array_equal(proof, want) and np.array_equal(rand, wrand)


def test_sharded_proof_host_transport_with_delays_and_reordering():
    """lock-order shake-out: four thread ranks on one GPU over a host transport that sleeps a random time before and after every
    exchange and lets the ranks enter in a random order, while a fifth thread keeps proving on the same GPU through the ordinary
    (pipelined, persistent-tail) path -- the per-device gate, the tail slot and the transport must never wait on each other in a cycle.
    Every proof equals the oracle's."""
    import random
    import time
    G, nv, nt, shapes = 4, 17, 4, [[0, 1, 2], [3, 3], [1]]
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 4400)
    ex = sharded.ThreadExchange(G)
    rnd = random.Random(5)

    def jitter_comm(rank):
        base = ex.comm
        inner_ar, inner_ag = {}, {}

        def allreduce(a):
            time.sleep(rnd.random() * 2e-3)
            parts = ex._exchange(rank, a)
            time.sleep(rnd.random() * 1e-3)
            tot = np.zeros_like(a)
            for x in parts:
                tot += x
            return tot

        def allgather(b):
            time.sleep(rnd.random() * 2e-3)
            got = np.concatenate(ex._exchange(rank, b))
            time.sleep(rnd.random() * 1e-3)
            return got

        return sharded.HostComm(rank, G, allreduce, allgather)

    out = [None] * G
    stop = threading.Event()
    side = {"n": 0, "err": None}

    def side_prover():
        try:
            _lib.check(sc.lib().sc_set_device(0))
            stabs = [cref.synth_table(4401, s, 1 << 14) for s in range(3)]
            scoefs = cref.synth_table(4401, 1000, 1)
            swant, _ = cref.ml_prove(H.desc_from(14, [[0, 1, 2]], stabs, scoefs), threads=1)
            poly, _ = H.hip_poly_from(14, [[0, 1, 2]], stabs, scoefs, device="cuda:0")
            while not stop.is_set():
                proof = sc.MLSumcheck.prove(poly)
                assert np.array_equal(np.stack([m.evaluations for m in proof]), swant)
                side["n"] += 1
        except Exception as e:  # noqa: BLE001
            import traceback
            side["err"] = f"{e}\n{traceback.format_exc()}"

    st = threading.Thread(target=side_prover)
    st.start()
    ts = [threading.Thread(target=_thread_rank, args=(r, G, nv, shapes, tabs, coefs, jitter_comm, 0, out, 3)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    stop.set()
    st.join(timeout=120)
    assert side["err"] is None, side["err"]
    assert side["n"] > 0
    for r in range(G):
        assert not isinstance(out[r], Exception) and out[r] is not None, out[r]
        for proof, rand in out[r]:
            assert np.array_equal(proof, want) and np.array_equal(rand, wrand), f"rank {r}"


@pytest.mark.parametrize("G,nv,nt,shapes,chunk", [
    (2, 14, 3, [[0, 1, 2]], 11),                              # 2 ranks x 4 chunks, merged kernel per chunk
    (4, 15, 4, [[0, 0, 1, 2, 3], [1, 2]], 11),                # 4 ranks x 4 chunks, per-product launches per chunk
])
def test_sharded_proof_with_streamed_shards(G, nv, nt, shapes, chunk):
    """out-of-core x multi-GPU: every rank keeps its shard of the tables in HOST memory and streams it through HBM chunk by chunk in
    rounds 1 and 2 (sc_prover_init_streamed on the shard), the sharded loop all-reducing the streamed rounds' lanes like any other round;
    whole proof vs the unsharded oracle, twice on the rewound handles (reference site of what this replaces: prover.rs:55-59, the deep
    copy of tables that do not fit)"""
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 4500 + nv)
    ex = sharded.ThreadExchange(G)
    out = [None] * G
    n_loc = (1 << nv) // G

    def run(rank):
        try:
            _lib.check(sc.lib().sc_set_device(0))
            eng = sharded.HipShardEngine(nv - (G.bit_length() - 1), shapes, coefs, [np.ascontiguousarray(t[rank * n_loc:(rank + 1) * n_loc]) for t in tabs],
                                         "cuda:0", streamed_chunk_log2=chunk)
            comm = ex.comm(rank)
            res = []
            for _ in range(2):
                eng.reset()
                res.append(sharded.prove_sharded_library(eng, comm, nv))
            comm.close()
            eng.close()
            out[rank] = res
        except Exception as e:  # noqa: BLE001
            import traceback
            out[rank] = RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")

    ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(G):
        assert not isinstance(out[r], Exception) and out[r] is not None, out[r]
        for proof, rand in out[r]:
            assert np.array_equal(proof, want), f"rank {r}"
            assert np.array_equal(rand, wrand), f"rank {r}"


def test_sharded_rounds_every_local_round_in_the_library():
    """sc_ml_prove_sharded stops sharding once the global instance is latency-bound (it gathers early), so most of a small test
    instance's rounds are replicated.  This drives ALL local rounds through the sharded loop (sc_ml_prove_sharded_rounds: per-round
    all-reduce, small and big rounds alike) on 4 thread ranks and checks their messages and challenges against the oracle."""
    G, nv, nt, shapes = 4, 20, 4, [[0, 1, 2], [3, 3], [1]]
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 4242)
    ex = sharded.ThreadExchange(G)
    out = [None] * G
    nl = nv - 2

    def run(rank):
        try:
            import torch
            _lib.check(sc.lib().sc_set_device(0))
            n_loc = (1 << nv) // G
            eng = sharded.HipShardEngine(nl, shapes, coefs, [t[rank * n_loc:(rank + 1) * n_loc] for t in tabs], "cuda:0", borrow=True)
            comm = ex.comm(rank)
            rng = sc.Blake2b512Rng.setup()
            lp, lr = np.empty((nl, eng.D, 4), dtype=np.uint64), np.empty((nl, 4), dtype=np.uint64)
            torch.cuda.current_stream().synchronize()
            _lib.check(sc.lib().sc_prover_set_stream(eng._h, None, 1))
            _lib.check(sc.lib().sc_ml_prove_sharded_rounds(eng._h, comm._h, rng._h, nv, nl, C.c_void_p(lp.ctypes.data), C.c_void_p(lr.ctypes.data)))
            comm.close()
            eng.close()
            out[rank] = (lp, lr)
        except Exception as e:  # noqa: BLE001
            import traceback
            out[rank] = RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")

    ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(G):
        assert not isinstance(out[r], Exception), out[r]
        assert np.array_equal(out[r][0], want[:nl]) and np.array_equal(out[r][1], wrand[:nl]), f"rank {r}"


def test_sharded_proof_world1_is_the_unsharded_proof():
    """one rank: no exchange, no tail; with the host transport and (next test) with RCCL"""
    nv, shapes, nt = 18, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 66)
    out = [None]
    _thread_rank(0, 1, nv, shapes, tabs, coefs, lambda r: sharded.HostComm(0, 1, lambda a: a, lambda b: b), 0, out)
    assert not isinstance(out[0], Exception), out[0]
    for proof, rand in out[0]:
        assert np.array_equal(proof, want) and np.array_equal(rand, wrand)


def test_sharded_proof_world1_rccl():
    nv, shapes, nt = 18, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 67)
    out = [None]

    def mk(_rank):
        c = sharded.NativeComm("cuda:0")
        _lib.check(sc.lib().sc_comm_selftest(c._h))
        return c
    _thread_rank(0, 1, nv, shapes, tabs, coefs, mk, 0, out)
    assert not isinstance(out[0], Exception), out[0]
    for proof, rand in out[0]:
        assert np.array_equal(proof, want) and np.array_equal(rand, wrand)


def _comm_probe(c, kind, G):
    """sc_comm_info and sc_comm_exchange_bench on a live communicator"""
    r, n, k = C.c_int(), C.c_int(), C.c_int()
    _lib.check(sc.lib().sc_comm_info(c._h, C.byref(r), C.byref(n), C.byref(k)))
    assert (n.value, k.value & 0xff) == (G, kind) and 0 <= r.value < G
    # RCCL: direct publication (the all-reduce delivers tagged lanes into the host-mapped page) is probed at init and holds on this box;
    # the other communicators never carry the flag
    assert bool(k.value & 0x100) == (kind == 1 and _lib.get_policy("rccl_direct") != 0), k.value
    mean, mn = C.c_double(), C.c_double()
    _lib.check(sc.lib().sc_comm_exchange_bench(c._h, 40, 20, C.byref(mean), C.byref(mn)))  # 40 words = a degree-4 message
    assert 0 < mn.value <= mean.value
    assert sc.lib().sc_comm_exchange_bench(c._h, 65, 1, C.byref(mean), None) == _lib.SC_ERR_BAD_ARG  # more words than an inbox slot holds
    return r.value


def test_comm_info_and_exchange_bench_on_every_kind_of_communicator():
    """the measurement hooks bench.py's N > 1 line uses (config.ranks_seen / config.exchange), on one GPU: RCCL with one rank, the
    peer-to-peer communicator and the host transport with thread ranks; the publish timeout is a process-wide setting"""
    _lib.check(sc.lib().sc_set_device(0))
    c = sharded.NativeComm("cuda:0")
    assert _comm_probe(c, 1, 1) == 0
    c.close()
    for kind in (3, 2):
        G = 4
        _P2P_GROUP[0] += 1
        gid = _P2P_GROUP[0]
        ex = sharded.ThreadExchange(G)
        out = [None] * G

        def work(rank):
            try:
                _lib.check(sc.lib().sc_set_device(0))
                c = sharded.P2PComm(gid, rank, G, "cuda:0") if kind == 3 else ex.comm(rank)
                out[rank] = _comm_probe(c, kind, G)
                c.close()
            except Exception as e:  # noqa: BLE001
                out[rank] = e
        ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
        assert out == list(range(G)), out
    assert sc.lib().sc_set_publish_timeout_ms(5000) == 0 and sc.lib().sc_set_publish_timeout_ms(0) == 0  # (0 restores the default)


def test_peer_to_peer_refuses_messages_longer_than_an_inbox_slot_before_touching_the_handle():
    """a product of 8 multiplicands has 9 evaluations = 72 words > 64.  sc_ml_prove_sharded over a p2p communicator must
    refuse it up front -- SC_ERR_BAD_ARG, the handle still at round 0 and usable -- not fail mid-proof"""
    nv, nt, shapes = 6, 8, [[0, 1, 2, 3, 4, 5, 6, 7]]
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 71)
    _P2P_GROUP[0] += 1
    gid = _P2P_GROUP[0]
    G, out = 2, [None, None]

    ex = sharded.ThreadExchange(G)

    def work(rank):
        try:
            _lib.check(sc.lib().sc_set_device(0))
            n_loc = (1 << nv) // G
            eng = sharded.HipShardEngine(nv - 1, shapes, coefs, [t[rank * n_loc:(rank + 1) * n_loc] for t in tabs], "cuda:0", borrow=True)
            c = sharded.P2PComm(gid, rank, G, "cuda:0")
            proof, rand = np.empty((nv, 9, 4), np.uint64), np.empty((nv, 4), np.uint64)
            rc = sc.lib().sc_ml_prove_sharded(eng._h, c._h, None, nv, C.c_void_p(proof.ctypes.data), C.c_void_p(rand.ctypes.data))
            msg = sc.lib().sc_last_error().decode()
            rnd = C.c_uint32(99)
            _lib.check(sc.lib().sc_prover_state(eng._h, None, None, None, C.byref(rnd)))
            c.close()
            hc = ex.comm(rank)
            got = sharded.prove_sharded_library(eng, hc, nv)
            hc.close()
            eng.close()
            out[rank] = (rc, msg, rnd.value, got)
        except Exception as e:  # noqa: BLE001
            import traceback
            out[rank] = RuntimeError(f"{e}\n{traceback.format_exc()}")

    ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    for r in range(G):
        assert not isinstance(out[r], Exception) and out[r] is not None, out[r]
        rc, msg, rnd, (proof, rand) = out[r]
        assert rc == _lib.SC_ERR_BAD_ARG and "at most 8 evaluations" in msg and rnd == 0
        assert np.array_equal(proof, want) and np.array_equal(rand, wrand)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _proc_rank(rank, world, port, backend, nv, shapes, nt, seed, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    device = rank if backend == "nccl" else 0
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tabs = [cref.synth_table(seed, s, 1 << nv) for s in range(nt)]
        coefs = cref.synth_table(seed, 1000, len(shapes))
        out = [None] * world
        mk = (lambda r: sharded.NativeComm(f"cuda:{device}")) if backend == "nccl" else (lambda r: sharded.HostComm.over_torch_distributed())
        _thread_rank(rank, world, nv, shapes, tabs, coefs, mk, device, out)
        q.put((rank, out[rank] if not isinstance(out[rank], Exception) else repr(out[rank])))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_procs(world, backend, nv, shapes, nt, seed):
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_proc_rank, args=(r, world, port, backend, nv, shapes, nt, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _, _, want, wrand = _oracle(nv, shapes, nt, seed)
    for r in range(world):
        assert not isinstance(res[r], str), res[r]
        for proof, rand in res[r]:
            assert np.array_equal(proof, want), f"rank {r}"
            assert np.array_equal(rand, wrand), f"rank {r}"


def test_sharded_proof_two_processes_one_gpu_over_gloo():
    """one process per rank as under torchrun; both ranks on this box's GPU, the library's host transport over gloo"""
    _run_procs(2, "gloo", 15, [[0, 1, 2], [3, 3], [1]], 4, 93)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_proof_rccl_one_process_per_gpu(world):
    """the production multi-GPU path: one process per GPU, RCCL all-reduce / all-gather on the provers' streams"""
    if _n_gpus() < world:
        pytest.skip(f"needs {world} visible GPUs (RCCL refuses two ranks on one device)")
    _run_procs(world, "nccl", 21, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10, 95)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_proof_rccl_one_thread_per_gpu(world):
    """single process, one thread per GPU, RCCL communicator created from the threads (ncclCommInitRank per thread)"""
    if _n_gpus() < world:
        pytest.skip(f"needs {world} visible GPUs")
    nv, shapes, nt = 21, [[0, 1, 2]], 3
    tabs, coefs, want, wrand = _oracle(nv, shapes, nt, 96)
    idb = (C.c_uint8 * 128)()
    _lib.check(sc.lib().sc_comm_unique_id(C.cast(idb, C.c_void_p)))

    class _Comm:
        def __init__(self, rank):
            self._h = C.c_void_p()
            _lib.check(sc.lib().sc_comm_init(C.cast(idb, C.c_void_p), rank, world, C.byref(self._h)))

        def close(self):
            sc.lib().sc_comm_free(self._h)

    out = [None] * world
    ts = [threading.Thread(target=lambda r=r: _thread_rank(r, world, nv, shapes, tabs, coefs, _Comm, r, out)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(world):
        assert not isinstance(out[r], Exception), out[r]
        for proof, rand in out[r]:
            assert np.array_equal(proof, want) and np.array_equal(rand, wrand)


# ---- f4: sharded GKR initialisation -----------------------------------------------------------------------------------------------
def _gkr_inputs(dim, seed):
    rng = np.random.default_rng(seed)
    n = 1 << dim
    idx = np.unique((rng.integers(0, 1 << dim, size=3 * n, dtype=np.uint64)) | (rng.integers(0, 1 << dim, size=3 * n, dtype=np.uint64) << np.uint64(dim))
                    | (rng.integers(0, 4, size=3 * n, dtype=np.uint64) << np.uint64(2 * dim)))[: 2 * n]
    return idx, cref.synth_table(seed, 1, idx.shape[0]), cref.synth_table(seed, 3, n), cref.synth_table(seed, 4, dim), cref.synth_table(seed, 5, dim)


@pytest.mark.parametrize("G,dim,strided", [(1, 9, False), (2, 10, True), (4, 12, False), (8, 7, True)])
def test_sharded_gkr_initialisation_logical_shards(G, dim, strided):
    """G logical shards of f1's non-zeros on one GPU, the caller-driven form: every shard's lanes (sc_gkr_phase_*_sharded in lanes
    mode), summed as integers, folded on the device (sc_wide_reduce_table) == the unsharded oracle tables"""
    from sumcheck_amd import sharded_gkr
    idx, vals, f3, g, u = _gkr_inputs(dim, 500 + dim)
    wh, wi, wv = cref.gkr_phase_one(idx, vals, dim, f3, g)
    wgu = cref.gkr_phase_two(wi, wv, dim, u)
    eng = sharded_gkr.HipGkrEngine()
    parts = [slice(r, None, G) for r in range(G)] if strided else [slice(r * (len(idx) // G), (r + 1) * (len(idx) // G) if r + 1 < G else None) for r in range(G)]
    tot, locals_ = None, []
    for sl in parts:
        lanes, oi, ov = eng.phase_one_partial(np.ascontiguousarray(idx[sl]), np.ascontiguousarray(vals[sl]), dim, f3, g)
        tot = lanes if tot is None else tot + lanes
        locals_.append((oi, ov))
    assert np.array_equal(eng.fold(tot), wh)
    # the distributed f1(g,.,.): per key, the sum over shards is the oracle's value (and every key the oracle has occurs somewhere)
    acc = {}
    for oi, ov in locals_:
        for k, v in zip(oi.tolist(), field.to_ints(ov)):
            acc[k] = (acc.get(k, 0) + v) % field.P
    assert sorted(acc) == wi.tolist() and [acc[k] for k in wi.tolist()] == field.to_ints(wv)
    tot2 = None
    for oi, ov in locals_:
        lanes = eng.phase_two_partial(oi, ov, dim, u)
        tot2 = lanes if tot2 is None else tot2 + lanes
    assert np.array_equal(eng.fold(tot2), wgu)


@pytest.mark.parametrize("G", [2, 4])
def test_sharded_gkr_initialisation_library_collective(G):
    """the library-driven form: one thread per rank, sc_gkr_phase_one_sharded / _two_sharded over an sc_comm (host transport
    between the threads; RCCL takes the same path on distinct GPUs): every rank ends with the complete tables"""
    from sumcheck_amd import sharded_gkr
    dim = 11
    idx, vals, f3, g, u = _gkr_inputs(dim, 77)
    wh, wi, wv = cref.gkr_phase_one(idx, vals, dim, f3, g)
    wgu = cref.gkr_phase_two(wi, wv, dim, u)
    ex = sharded.ThreadExchange(G)
    out = [None] * G

    def run(rank):
        try:
            _lib.check(sc.lib().sc_set_device(0))
            comm = ex.comm(rank)
            f1 = sc.SparseMultilinearExtension(3 * dim, np.ascontiguousarray(idx[rank::G]), np.ascontiguousarray(vals[rank::G]))
            h_g, f1_g = sharded_gkr.initialize_phase_one_sharded(comm, f1, sc.DenseMultilinearExtension(dim, f3), g)
            f1_gu = sharded_gkr.initialize_phase_two_sharded(comm, f1_g, u)
            comm.close()
            out[rank] = (h_g.evaluations, f1_gu.evaluations)
        except Exception as e:  # noqa: BLE001
            import traceback
            out[rank] = RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")

    ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(G):
        assert not isinstance(out[r], Exception), out[r]
        assert np.array_equal(out[r][0], wh) and np.array_equal(out[r][1], wgu), f"rank {r}"


@pytest.mark.parametrize("G,dim,transport", [(2, 10, "host"), (4, 12, "host"), (4, 11, "p2p"), (8, 9, "p2p")])
def test_sharded_gkr_round_sumcheck_end_to_end(G, dim, transport):
    """sc_gkr_prove_sharded: GKRRoundSumcheck::prove with f1's non-zeros spread over G thread ranks AND both sumcheck phases sharded
    (each rank proves over its high-bit slice of the phase's two tables, per-round all-reduce, early gather, replicated tail), over the
    host transport and over the peer-to-peer communicator.  Every rank's proof, u and v equal the unsharded oracle's
    (reference gkr_round_sumcheck/mod.rs:93-139; shape of gkr_round_sumcheck/test.rs:24-88)."""
    from sumcheck_amd import sharded_gkr
    idx, vals, f3, g, _ = _gkr_inputs(dim, 900 + dim)
    f2 = cref.synth_table(900 + dim, 6, 1 << dim)
    want, wuv = cref.gkr_prove(idx, vals, dim, f2, f3, g, threads=4)
    ex = sharded.ThreadExchange(G)
    _P2P_GROUP[0] += 1
    gid = _P2P_GROUP[0]
    out = [None] * G

    def run(rank):
        try:
            _lib.check(sc.lib().sc_set_device(0))
            comm = ex.comm(rank) if transport == "host" else sharded.P2PComm(gid, rank, G, "cuda:0")
            f1 = sc.SparseMultilinearExtension(3 * dim, np.ascontiguousarray(idx[rank::G]), np.ascontiguousarray(vals[rank::G]))
            res = []
            for _ in range(2):
                res.append(sharded_gkr.prove_sharded(comm, sc.Blake2b512Rng.setup(), f1, sc.DenseMultilinearExtension(dim, f2),
                                                     sc.DenseMultilinearExtension(dim, f3), g))
            comm.close()
            out[rank] = res
        except Exception as e:  # noqa: BLE001
            import traceback
            out[rank] = RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")

    ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(G):
        assert not isinstance(out[r], Exception) and out[r] is not None, out[r]
        for m1, m2, u, v in out[r]:
            assert np.array_equal(m1, want[0]) and np.array_equal(m2, want[1]), f"rank {r}"
            assert np.array_equal(u, wuv[0]) and np.array_equal(v, wuv[1]), f"rank {r}"
