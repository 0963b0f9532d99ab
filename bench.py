#!/usr/bin/env python3
"""bench.py -- MLSumcheck prover field-ops/s (BLS12-381 Fr) on N MI355X (BASELINE.json's metric).

A "step" is one complete MLSumcheck::prove over HBM-resident synthetic tables: every round's fused bind+sum
kernels, the round polynomial landing on the host, the host Fiat-Shamir hash and the challenge going back.

Workloads (named in config.workload):
  --config 3 (default)  BASELINE config 3, the configuration the metric is quoted on: products [0,1,2,3],[4,5,6],[7,8],[9] over
                        10 tables (5 GiB at nv=24).
      --scaling strong (default)  the metric as worded, "nv=24 at 1/2/4/8 MI355X": the SAME nv=24 instance split N ways by the
                                  high index bits (2^24/N entries of every table per GPU).
      --scaling weak              every GPU holds a config-3-sized shard: global nv = 24 + log2 N.
  --config 4            BASELINE config 4: one product of 3 tables, nv=28 (24 GiB), sharded N ways (nv_local = 28 - log2 N; N=1 holds
                        the whole instance, 38 GiB with the bound-table buffers).
At N>1 the proof is one sc_ml_prove_sharded call per rank (local rounds with one integer all-reduce of the round polynomial per
round, bind + all-gather, replicated tail rounds).

value = field_ops(global instance) / t, field_ops as executed by the reference algorithm:
  (2^nv - 1) * sum_k (2 m_k D + m_k + D) + 3 U (2^nv - 2).

Launch.  `python bench.py --gpus N` is all it takes: with N > 1 and no launcher in the environment (WORLD_SIZE unset) the command
starts its own ranks -- one process per GPU through `python -m torch.distributed.run` on 127.0.0.1 (RCCL inside the library), and if
that cannot be started, N thread ranks of this process over the library's peer-to-peer communicator (sc_comm_init_p2p: no collective
library at all).  `--launcher processes|threads` forces one.  Under an external launcher
(`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) it is one of the ranks, as before.  Rank 0 prints ONE JSON
line, the last line of stdout.
"""
import argparse
import ctypes as C
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

# the CPU leg's OpenMP threads: one per core, pinned (set before any OpenMP runtime is loaded; a caller's own settings win)
OMP_DEFAULTS_SET_HERE = [k for k in ("OMP_PROC_BIND", "OMP_PLACES") if k not in os.environ]
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")
# The CPUs this command may use, taken BEFORE any OpenMP runtime exists: with OMP_PROC_BIND the first runtime that loads binds this
# process's main thread to one core, and a child process inherits that one-core mask -- the ranks `--gpus N` starts would then run
# rank 0's CPU leg on a single core (seen: 1.2e8 instead of 8.9e8 field-ops/s).  self_launch() hands the launcher this mask back and
# keeps the two variables above out of the LAUNCHER's environment (it loads an OpenMP runtime too, and the ranks are its children);
# every rank is this file again and sets them for itself.
CPUS_AT_START = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x5C20241008
C3_SHAPES = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]
C4_SHAPES = [[0, 1, 2]]
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FE_MUL_CEILING_PER_S = 160e9  # the carry-free product's in-kernel rate, chip-wide (profiles/r1_modmul_ceiling.txt, tools/modmul_bench.py)
BIG_ROUND_MIN_PAIRS_LOG2 = 15  # rounds with at least 2^15 pairs (more than kernels.h's kSmallRoundPairs = 2^14) run the merged big-round kernel


def field_ops(nv, shapes, n_tables):
    D = max(len(s) for s in shapes) + 1
    return ((1 << nv) - 1) * sum(2 * len(s) * D + len(s) + D for s in shapes) + 3 * n_tables * ((1 << nv) - 2)


def reference_muls(nv, shapes, n_tables):
    """field multiplications of the reference algorithm per proof (prover.rs:110-148: every product at every point; fix_variables)"""
    D = max(len(s) for s in shapes) + 1
    return ((1 << nv) - 1) * sum(len(s) * D for s in shapes) + n_tables * ((1 << nv) - 2)


def executed_products(nv_local, shapes, n_tables):
    """Montgomery products the kernels execute per proof on one shard, in units of one general product (153 multiply-adds;
    a constant-multiplier bind is 97 of them): the product tree at M+1 nodes (0 / 3 / 7 / 11 products for 1..4 multiplicands, one less
    per product in the big binding rounds that take node 1 from the claim identity), two binds per table and pair in every binding
    round; the latency-bound rounds evaluate every (product, node) combination on its own (M - 1 products each, general-product binds)."""
    tree = {1: 0, 2: 3, 3: 7, 4: 11}
    total = 0.0
    for i in range(1, nv_local + 1):
        pairs = 1 << (nv_local - i)
        big = (nv_local - i) >= BIG_ROUND_MIN_PAIRS_LOG2 and all(len(s) <= 4 for s in shapes)
        if big:
            per_pair = sum(tree[len(s)] - (1 if (i >= 2 and len(s) >= 2) else 0) for s in shapes)
            binds = (2 * n_tables * 97.0 / 153.0) if i >= 2 else 0.0
        else:
            per_pair = sum((len(s) + 1) * (len(s) - 1) for s in shapes)
            binds = 2 * n_tables if i >= 2 else 0.0
        total += pairs * (per_pair + binds)
    return total


def algorithmic_bytes(nv, n_tables):
    """compulsory HBM traffic of the fused schedule: round 1 reads every table once; round i>=2 reads T_{i-1}
    once and writes T_i once: 32 * U * (4 * 2^nv - 6)"""
    return 32 * n_tables * (4 * (1 << nv) - 6)


def round_bytes(nv_local, n_tables, i):
    """algorithmic bytes of round i (1-based) of a shard of 2^nv_local entries per table"""
    if i == 1:
        return 32 * n_tables * (1 << nv_local)
    return 32 * n_tables * ((1 << (nv_local - i + 2)) + (1 << (nv_local - i + 1)))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def file_sha(*paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def source_shas():
    """content hashes that tie an off-line measurement (profiles/hbm_traffic_latest.json) to the tree that is running: the kernel sources
    and this file (a GPU box has no .git)"""
    cs = os.path.join(ROOT, "sumcheck_amd", "csrc")
    return {"csrc_sha": file_sha(*[os.path.join(cs, f) for f in sorted(os.listdir(cs)) if os.path.isfile(os.path.join(cs, f))]),
            "bench_py_sha": file_sha(os.path.abspath(__file__))}


def cpu_baseline(shapes, n_tables, nv_full=24, budget_s=12.0):
    """The CPU restatement of the reference algorithm (oracle/oracle.c: the loop nest of prover.rs:110-148 in rayon-shaped OpenMP,
    >= 1024-point chunks, bind parallel across tables only as prover.rs:87) timed on this host's cores on a bounded sample of the
    same workload, in the three flavours SURVEY 8(d) asks for: all cores (the reported value), one thread, and all cores with the
    bind also parallel inside a table ("improved": NOT what the reference does).

    Like for like with the GPU clock, which starts with the tables resident in HBM: `value` is the prove_round loop alone (bind +
    sums, the per-phase split is reported); prover_init's deep copy of the tables (prover.rs:55-59) is timed separately and only
    enters `whole_prove`.  The all-cores proof of the full-size instance is kept: bench.py compares the GPU proofs with it."""
    from oracle import cref
    threads = cref.max_threads()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("OMP_NUM_THREADS") == "1":
        # torch.distributed.run exports OMP_NUM_THREADS=1 to every rank it starts unless the caller set it.  The CPU leg is rank 0's
        # alone (the other ranks wait) and is meant to use the host's cores: the hardware threads, capped by the container's quota.
        q = cref.cpu_quota_cores()
        hw = len(CPUS_AT_START) if CPUS_AT_START else (os.cpu_count() or 1)  # (not sched_getaffinity now: the main thread is bound by then)
        threads = max(1, min(hw, q) if q else hw)
    coefs = cref.synth_table(SEED, 1000, len(shapes))

    def run(nv, nthreads, improved=False):
        tabs = [cref.synth_table(SEED, s, 1 << nv) for s in range(n_tables)]
        d = cref.PolyDesc(nv, [(coefs[k], s) for k, s in enumerate(shapes)], tabs)
        proof = None
        t0 = time.perf_counter()
        if improved:
            rng = cref.Rng()
            rng.feed_poly_info(d.max_multiplicands, nv)
            pr = cref.Prover(d, threads=nthreads, improved_fix=True)
            r = None
            for _ in range(nv):  # mod.rs:57-64
                m = pr.prove_round(r)
                rng.feed_prover_msg(m)
                r = rng.sample_fr()
            phases = pr.times()
            pr.close()
        else:
            proof, _ = cref.ml_prove(d, threads=nthreads)
            phases = cref.last_prove_times()
        return time.perf_counter() - t0, phases, proof

    def sized(nthreads, improved, budget):  # the largest nv <= nv_full whose run fits the budget, found by doubling
        nv = min(14 if nthreads == 1 else 18, nv_full)
        t, ph, proof = run(nv, nthreads, improved)
        while nv < nv_full and 2.3 * t < budget:
            nv += 1
            t, ph, proof = run(nv, nthreads, improved)
        ops = field_ops(nv, shapes, n_tables)
        rounds_s = ph[1] + ph[2]
        return {"value": ops / rounds_s, "unit": "field-ops/s", "cores": nthreads,
                "sample": f"same products at nv={nv} ({ops:.3e} field-ops; prove_round loop {rounds_s:.2f} s = bind {ph[1]:.2f} + sums {ph[2]:.2f}; "
                          f"prover_init copy {ph[0]:.2f} s; whole prove {t:.2f} s)",
                "phases_s": {"init_copy": ph[0], "bind": ph[1], "sums": ph[2], "whole_prove": t},
                "whole_prove": {"value": ops / t, "unit": "field-ops/s"}, "_nv": nv, "_proof": proof}

    allc = sized(threads, False, budget_s)
    one = sized(1, False, budget_s / 2)
    imp = sized(threads, True, budget_s)
    kept = (allc.pop("_nv"), allc.pop("_proof"))
    for d in (one, imp):
        d.pop("_nv"), d.pop("_proof")
    return {"value": allc["value"], "unit": "field-ops/s", "cores": threads, "kind": "port",
            "sample": allc["sample"] + f", OpenMP {threads} threads, OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}",
            "phases_s": allc["phases_s"], "whole_prove": allc["whole_prove"],
            "cpu_model": cpu_model(), "host_cores": os.cpu_count(), "cpu_quota_cores": cref.cpu_quota_cores(),
            "cpus_allowed": len(CPUS_AT_START) if CPUS_AT_START else None,  # this process's affinity mask when it started
            "one_thread": one, "all_cores_improved_bind": imp,
            "note": "port = oracle/oracle.c, a C restatement of the reference algorithm (the Rust reference cannot be built here). value = the "
                    "prove_round loop with the tables already copied (what the GPU clock covers); whole_prove adds prover_init's deep copy "
                    "(done by all threads in slices -- the reference clones serially). cores = the threads started: the host's hardware threads "
                    "capped by the container's CPU quota (cpu_quota_cores; more threads than the quota are throttled, not run). all_cores_improved_bind parallelises fix_variables "
                    "inside a table, which the reference does not"}, kept


# ---------------------------------------------------------------------------------------------------------------------------------
# Ranks.  A World is what the measurement needs from "the other ranks": a barrier, a max, an any, a gather of small arrays, and the
# library communicator the sharded proof runs over.
# ---------------------------------------------------------------------------------------------------------------------------------
class ProcWorld:
    """one process per GPU under torch.distributed (RCCL = backend "nccl"; SC_BENCH_ONE_GPU=1 -- tests -- puts every rank on GPU 0 and
    exchanges through gloo, because RCCL refuses two ranks on one device)"""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.one_gpu = os.environ.get("SC_BENCH_ONE_GPU") == "1"
        self.local_rank = 0 if self.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        self.launcher = ("processes (self-launched torch.distributed.run)" if os.environ.get("SC_BENCH_SELF_LAUNCHED") == "1"
                         else "processes (external launcher)") if self.world > 1 else "single process"
        self.dist = None
        # SC_BENCH_FORCE_SHARDED=1 with one rank (tests): everything an N > 1 RCCL run calls -- the "nccl" process group with its barrier,
        # device all-reduce and all-gather, the library's RCCL communicator, the sharded proof -- runs with a world of one
        self.force_dist = self.world == 1 and os.environ.get("SC_BENCH_FORCE_SHARDED") == "1" and not self.one_gpu
        if self.force_dist:
            self.launcher = "single process (the N > 1 code path forced: one-rank RCCL world)"

    def init(self, dev):
        if self.force_dist:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            s.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if self.world > 1 or self.force_dist:
            import datetime
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # (rank 0 arrives after its CPU leg: the others wait here, hence the generous timeout)
            if self.one_gpu:
                dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=30))
            self.dist = dist
            self._dev = "cpu" if self.one_gpu else dev

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_float(self, x):
        if not self.dist:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=self._dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def any(self, flag):
        return self.max_float(1.0 if flag else 0.0) > 0.0

    def gather(self, arr):
        """arr: small uint64 numpy array, same shape on every rank -> list over ranks"""
        if not self.dist:
            return [arr]
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr).view(np.int64).copy()).to(self._dev)
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [p.cpu().numpy().view(np.uint64).reshape(arr.shape) for p in parts]

    def make_comm(self, dev):
        from sumcheck_amd import sharded
        if self.one_gpu:
            return sharded.HostComm.over_torch_distributed(), "library+host-transport(gloo)"
        return sharded.NativeComm(dev), "library+rccl"

    def finish(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()


class ThreadWorld:
    """N thread ranks of ONE process, one GPU each (how a Rust host would drive the library: sc_set_device per thread), over the
    library's peer-to-peer communicator -- no torch.distributed, no RCCL.  SC_BENCH_ONE_GPU=1 (tests) puts every rank on GPU 0."""
    _GROUP = 0xBE7C0000

    class Shared:
        def __init__(self, n):
            self.n = n
            self.bar = threading.Barrier(n)
            self.slots = [None] * n
            self.group = ThreadWorld._GROUP + (os.getpid() & 0xffff)

    def __init__(self, shared, rank):
        self.sh = shared
        self.world = shared.n
        self.rank = rank
        self.one_gpu = os.environ.get("SC_BENCH_ONE_GPU") == "1"
        self.local_rank = 0 if self.one_gpu else rank
        self.launcher = "threads (one process, sc_comm_init_p2p)"

    def init(self, dev):
        pass

    def barrier(self):
        self.sh.bar.wait(timeout=3600)

    def _exchange(self, v):
        self.sh.slots[self.rank] = v
        self.sh.bar.wait(timeout=3600)
        got = list(self.sh.slots)
        self.sh.bar.wait(timeout=3600)
        return got

    def max_float(self, x):
        return max(self._exchange(x))

    def any(self, flag):
        return any(self._exchange(bool(flag)))

    def gather(self, arr):
        return self._exchange(np.array(arr, copy=True))

    def make_comm(self, dev):
        from sumcheck_amd import sharded
        return sharded.P2PComm(self.sh.group, self.rank, self.world, dev), "library+p2p"

    def finish(self):
        self.sh.bar.wait(timeout=3600)


# ---- the scaling model: written down BEFORE a multi-GPU node exists, so that the first N > 1 line tests it ----------------
# One-GPU whole-proof times of the shard shapes, ms (profiles/r5n_single_gpu_shards.json: `python bench.py --config C --nv n` on one
# MI355X, round 5).  A rank of an N-GPU proof holds nv - log2 N variables per table.
T1_MS = {3: {24: 5.00, 23: 2.77, 22: 1.63, 21: 1.03, 20: 0.71}, 4: {28: 23.0, 27: 12.28, 26: 6.28, 25: 3.33}}
EXCHANGE_ASSUMED_US = {"rccl": {2: 12.0, 4: 15.0, 8: 20.0, 16: 25.0}, "p2p": {2: 6.0, 4: 7.0, 8: 8.0, 16: 10.0}}  # ASSUMED (no two-GPU box seen yet)
GATHER_ASSUMED_GBPS, GATHER_ASSUMED_LATENCY_US = 50.0, 30.0  # all-gather over xGMI: per-rank receive rate, plus bind / launch / tail-reset latency
REPLICATED_ROUND_US = {14: 29.0, 13: 20.0, 12: 16.0, 11: 15.5}  # a latency-bound round by log2(pairs), measured on one GPU out of LDS


def t1_ms(config, nv):
    """one-GPU proof time of an nv-variable instance of the config's shape: the table, else scaled from the nearest entry (bytes halve per variable)"""
    tab = T1_MS[config]
    if nv in tab:
        return tab[nv], "measured"
    near = min(tab, key=lambda q: abs(q - nv))
    fixed = 0.45 if config == 3 else 0.5  # the latency-bound rounds' share, ms: does not scale
    return fixed + (tab[near] - fixed) * 2.0 ** (nv - near), f"scaled from nv={near}"


def predict_ms(config, nv_total, world, U, comm_kind, exchange_us=None):
    """predicted ms per proof at `world` GPUs: the shard's own one-GPU time + one exchange per sharded round + the early gather + the
    log2(world) extra replicated rounds (the replicated tail starts at 2^14 pairs whatever the shard size)"""
    k = world.bit_length() - 1
    nv_local = nv_total - k
    m = min(max(15 - k, 0), nv_local - 1) if k else 0
    nl = nv_local - m
    t1, t1_src = t1_ms(config, nv_local)
    if world == 1:
        return {"predicted_ms_per_step": t1, "t1_ms": t1, "t1_source": t1_src}
    kind = "p2p" if comm_kind == "p2p" else "rccl"
    x = EXCHANGE_ASSUMED_US[kind].get(world, 25.0)
    gather_bytes = U * (1 << m) * 32 * (world - 1)  # received per rank
    gather_us = gather_bytes / (GATHER_ASSUMED_GBPS * 1e3) + GATHER_ASSUMED_LATENCY_US
    extra_us = sum(REPLICATED_ROUND_US.get(14 - j, 25.0) for j in range(k))
    out = {"model": "T1(nv per GPU) + sharded_rounds * exchange + gather + log2(N) extra replicated rounds",
           "t1_ms": t1, "t1_source": t1_src + " (profiles/r5n_single_gpu_shards.json)", "sharded_rounds": nl, "replicated_rounds": m + k,
           "exchange_assumed_us": x, "gather_bytes_received_per_rank": gather_bytes, "gather_assumed_us": gather_us,
           "gather_assumption": f"{GATHER_ASSUMED_GBPS:.0f} GB/s per rank + {GATHER_ASSUMED_LATENCY_US:.0f} us", "extra_replicated_rounds_us": extra_us,
           "predicted_ms_per_step": t1 + (nl * x + gather_us + extra_us) * 1e-3}
    if exchange_us:  # the same with the exchange this run measured on its communicator
        out["predicted_ms_per_step_with_measured_exchange"] = t1 + (nl * exchange_us + gather_us + extra_us) * 1e-3
    return out


def comm_info(sc, comm):
    r, n, k = C.c_int(), C.c_int(), C.c_int()
    from sumcheck_amd import _lib
    _lib.check(sc.lib().sc_comm_info(comm._h, C.byref(r), C.byref(n), C.byref(k)))
    return r.value, n.value, {1: "rccl", 2: "host-transport", 3: "p2p"}.get(k.value & 0xff, str(k.value)), bool(k.value & 0x100)


def end_to_end(sc, _lib, torch, tables, shapes, coefs, nv, dev, want, reps=5):
    """SURVEY 8d's t_end_to_end: the path a Rust caller takes -- HOST tables in, proof out (`MLSumcheck::prove(&poly)`, mod.rs:42-53, whose
    prover_init deep copy, prover.rs:55-59, is the host-to-device copy here).  Measured AFTER the timed region, never part of `value`:
    the bare copy of the same bytes (pinned and pageable), and whole one-shot proofs from host tables with the staged initialisation
    (round 1 under the copy; protocol.hip staged_copy_and_round1) and without it.  Every proof is compared with the timed region's."""
    def wall(f):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3, r

    out = {"bytes": int(sum(t.numel() * 8 for t in tables)), "reps": reps,
           "what": "after the timed region; one-shot sc_ml_prove over HOST tables (the reference's calling convention), best of reps; h2d_ms = the bare hipMemcpy of the same bytes"}
    host = [t.cpu() for t in tables]  # pageable
    scratch = torch.empty_like(tables[0])
    ok = True
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            host = [h.pin_memory() for h in host]

        def bare():
            for h in host:
                scratch.copy_(h, non_blocking=True)
        h2d = min(wall(bare)[0] for _ in range(reps))
        mles = [sc.DenseMultilinearExtension(nv, h) for h in host]
        poly = sc.ListOfProductsOfPolynomials(nv)
        for kk, sh in enumerate(shapes):
            poly.add_product([mles[i] for i in sh], coefs[kk])
        res = {"h2d_ms": h2d, "h2d_GBps": out["bytes"] / h2d / 1e6}
        for staged in (1, 0):
            with _lib.policy(staged_init=staged):
                ts = []
                for _ in range(reps + 1):  # (the first call builds the prover the library keeps for this shape)
                    ms, proof = wall(lambda: sc.MLSumcheck.prove(poly))
                    ts.append(ms)
                    ok = ok and bool(np.array_equal(np.stack([m.evaluations for m in proof]).reshape(want.shape), want))
            key = "staged" if staged else "copy_then_prove"
            res[key + "_total_ms"] = min(ts[1:])
            res[key + "_first_call_ms"] = ts[0]
        res["total_ms"] = res["staged_total_ms"]
        res["prove_beyond_h2d_ms"] = res["staged_total_ms"] - h2d
        out[kind] = res
    out["proofs_equal_timed_region"] = ok
    return out


def run_rank(args, W, result):
    """the measurement on one rank; rank 0 leaves the JSON line's dictionary in result['line'] and every rank its exit status"""
    import torch
    import sumcheck_amd as sc
    from sumcheck_amd import _lib, sharded

    world, rank, local_rank, one_gpu = W.world, W.rank, W.local_rank, W.one_gpu
    assert world & (world - 1) == 0, "the shard count must be a power of two"
    k = world.bit_length() - 1
    if args.config == 4:
        shapes, U, nv_total, scaling = C4_SHAPES, 3, 28, "strong"
    else:
        shapes, U = C3_SHAPES, 10
        scaling = args.scaling
        nv_total = 24 + (k if scaling == "weak" else 0)
    if args.nv:
        nv_total = args.nv
    nv_local = nv_total - k
    n_loc = 1 << nv_local
    D = max(len(s) for s in shapes) + 1

    # The CPU leg runs FIRST, on rank 0 (at every N: a line without it is unmeasured), so that the GPU leg is the last thing this
    # command does and an outside sampler of GPU activity sees it.  Its sample is the SAME products at up to nv = 24; when that is the
    # instance the GPUs prove (config 3, strong scaling: the metric as worded) its proof is kept and the GPU proofs are compared with it.
    force_sharded = os.environ.get("SC_BENCH_FORCE_SHARDED") == "1"  # exercise the N>1 code path on one GPU (tests)
    cpu, cpu_proof_nv, cpu_proof = None, 0, None
    if rank == 0 and not args.no_cpu_baseline:
        try:
            cpu, (cpu_proof_nv, cpu_proof) = cpu_baseline(shapes, U, nv_full=min(nv_total, 24))
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
            cpu = {"value": None, "unit": "field-ops/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    for kv in args.policy:  # (process-wide: thread ranks set the same values again, harmlessly)
        key, _, val = kv.partition("=")
        _lib.set_policy(key.strip(), int(val))
    torch.cuda.set_device(local_rank)  # (thread-local, like sc_set_device)
    dev = torch.device("cuda", local_rank)
    _lib.check(sc.lib().sc_set_device(local_rank))
    W.init(dev)

    # synthetic tables generated on the device; rank g holds entries [g*2^nv_local, (g+1)*2^nv_local) of every table
    tables = []
    for u in range(U):
        t = torch.empty((n_loc, 4), dtype=torch.int64, device=dev)
        _lib.check(sc.lib().sc_synth_table_device(SEED, u, rank * n_loc, n_loc, C.c_void_p(t.data_ptr())))
        tables.append(t)
    ct = torch.empty((len(shapes), 4), dtype=torch.int64, device=dev)
    _lib.check(sc.lib().sc_synth_table_device(SEED, 1000, 0, len(shapes), C.c_void_p(ct.data_ptr())))
    coefs = ct.cpu().numpy().view(np.uint64)
    torch.cuda.synchronize(dev)

    round_loop = "library"
    sharded_path = world > 1 or force_sharded
    exchange = None
    ranks_seen, comm_kind, direct_pub = 1, "none", False
    box = {"comm": None, "python": False, "why": "single GPU: no exchange"}
    if not sharded_path:
        mles = [sc.DenseMultilinearExtension(nv_local, t) for t in tables]
        poly = sc.ListOfProductsOfPolynomials(nv_local)
        for kk, sh in enumerate(shapes):
            poly.add_product([mles[i] for i in sh], coefs[kk])
        state = sc.IPForMLSumcheck.prover_init(poly, borrow=True)  # tables stay where they are: no copy
        handle = state._h

        def step():
            state.reset()
            return state.prove()
    else:
        engine = sharded.HipShardEngine(nv_local, shapes, coefs, tables, dev, borrow=True)
        handle = engine._h
        box["python"] = os.environ.get("SC_BENCH_PYTHON_ROUNDS") == "1"
        box["why"] = "SC_BENCH_PYTHON_ROUNDS=1" if box["python"] else ""
        if not box["python"]:
            try:  # the whole sharded proof inside the library: RCCL on the prover's stream, the peer-to-peer exchange kernel, or (one-GPU test mode) its host transport
                box["comm"], round_loop = W.make_comm(dev)
                # collective self-test BEFORE the warm-up: one all-reduce and one all-gather of known patterns, checked on every rank
                _lib.check(sc.lib().sc_comm_selftest(box["comm"]._h))
                box["why"] = "sc_comm_selftest passed on every rank"
                _, ranks_seen, comm_kind, direct_pub = comm_info(sc, box["comm"])
            except Exception as e:
                box["why"] = f"in-library communicator unavailable or failed its self-test: {e}"
                box["python"] = True
        if world > 1 and W.any(box["python"]) and not box["python"]:  # every rank takes the same path
            box["python"], box["why"] = True, "another rank's communicator failed its self-test"
        if box["python"] and not isinstance(W, ProcWorld):
            raise RuntimeError(f"thread ranks have no torch.distributed fallback: {box['why']}")
        log(f"[bench] rank {rank}: round loop = {'torch.distributed (fallback)' if box['python'] else round_loop} -- {box['why']}")
        if not box["python"]:
            # the per-round exchange on its own: back-to-back all-reduces of one round message (D x 8 lanes = 320 bytes for degree 4) in
            # the form a sharded round uses on this communicator, each waited for by the host like a round's
            # (a rank whose measurement raised must not skip a collective its peers enter: every rank agrees on the outcome first)
            iters = 100 if one_gpu else 1000
            us_mean, us_min = C.c_double(), C.c_double()
            err = None
            try:
                _lib.check(sc.lib().sc_comm_exchange_bench(box["comm"]._h, 8 * D, iters, C.byref(us_mean), C.byref(us_min)))
            except Exception as e:
                err = f"{type(e).__name__}: {e}"
            if W.any(err is not None):
                exchange = {"exchange_us": None, "reason": err or "the measurement failed on another rank"}
            else:
                exchange = {"exchange_us": W.max_float(us_mean.value), "exchange_us_min": us_min.value, "bytes": 64 * D, "iters": iters,
                            "publication": ("direct: ncclAllReduce delivers tagged lanes into the host-mapped page, no publish kernel" if direct_pub else
                                            "the exchange kernel publishes" if comm_kind == "p2p" else
                                            "the host adds the ranks' lanes (the caller's all-reduce function)" if comm_kind == "host-transport" else
                                            "publish kernel behind the all-reduce"),
                            "what": f"{iters} back-to-back all-reduces of one round message on the {comm_kind} communicator, host-waited like a round's (max over ranks of the mean)"}
        dcomm = sharded.DistComm() if isinstance(W, ProcWorld) else None
        tail_factory = sharded.TailEngines(shapes, coefs, dev)  # only the Python loop uses it

        def step():
            engine.reset()
            if not box["python"]:
                try:
                    return sharded.prove_sharded_library(engine, box["comm"], nv_total)[0]
                except Exception as e:  # same on every rank (collective failure): drop to the torch.distributed loop for good
                    if dcomm is None:
                        raise
                    log(f"[bench] in-library sharded proof failed ({e}); falling back to the torch.distributed round loop")
                    box["python"] = True
                    engine.reset()
            return sharded.prove_sharded([engine], dcomm, nv_total, max(len(s) for s in shapes), tail_factory)[0]

    def barrier():
        torch.cuda.synchronize(dev)
        W.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        proof = step()
    try:  # every rank: flush RCCL's NCCL_DEBUG=VERSION banner (C stdio) now, long before rank 0 prints the JSON line
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    # The dominant kernel's duration is measured live, with HIP events on the launch stream inside the timed region -- on a sample of
    # the steps only: the events are instrumentation (four records and a collect per big round, ~3 % of a proof: gpu_leg.ms_per_event_sampled_proof), and the other steps
    # run as a caller's proofs do.  At least 4 steps are sampled whatever --steps is (kernel durations repeat to a percent).  The sampled steps are part of the K timed steps.
    K = len(shapes)
    ms_acc, ln_acc, rounds_ms_acc, timed_steps = [0.0] * K, [0] * K, 0.0, 0
    rms_acc, rln_acc = [0.0] * nv_local, [0] * nv_local
    ms = (C.c_double * K)()
    ln = (C.c_uint64 * K)()
    rms = (C.c_double * nv_local)()
    rln = (C.c_uint64 * nv_local)()
    rounds_ms = C.c_double()
    every = max(1, min(args.time_every, args.steps // 4))  # (a sampled proof costs ~0.15 ms more than the others: 36 event records and 9 collects)
    barrier()
    t0 = time.perf_counter()
    step_s = []  # a step returns when its last round's message is on the host, so per-step wall times cost nothing extra
    for i in range(args.steps):
        sampled = i % every == 0
        if sampled:
            _lib.check(sc.lib().sc_prover_set_timing(handle, 1))
        ts = time.perf_counter()
        proof = step()
        step_s.append(time.perf_counter() - ts)
        if sampled:
            _lib.check(sc.lib().sc_prover_get_timing(handle, ms, ln, C.byref(rounds_ms)))
            _lib.check(sc.lib().sc_prover_get_round_timing(handle, rms, rln))
            _lib.check(sc.lib().sc_prover_set_timing(handle, 0))
            for q in range(K):
                ms_acc[q] += ms[q]
                ln_acc[q] += ln[q]
            for q in range(nv_local):
                rms_acc[q] += rms[q]
                rln_acc[q] += rln[q]
            rounds_ms_acc += rounds_ms.value
            timed_steps += 1
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = W.max_float(elapsed)

    # ---- parity: the line certifies itself ---------------------------------------------------------------------------------------
    # Where the CPU leg proved THIS instance (N = 1; N > 1 with config 3 strong scaling: the tables are seed-defined, so the sharded
    # instance IS the CPU leg's instance): the last TIMED proof and one more proof after the timed region are compared, message by
    # message, with the CPU proof, on every rank.  Elsewhere (weak scaling, config 4: no CPU proof of the instance exists) the proof
    # certifies itself the way the reference's own tests do (ml_sumcheck/test.rs:71-74): the verifier accepts every round and its
    # final oracle query is answered from the sharded tables.
    have_cpu_proof = cpu_proof is not None and cpu_proof_nv == nv_total
    all_have = W.any(have_cpu_proof)  # rank 0 decides (it alone ran the CPU leg)
    parity = {"vs": None, "ok": None, "reason": "no CPU proof of this instance in this run (--no-cpu-baseline, or the CPU sample stopped below the full size)"}
    timed_proof = np.asarray(proof, dtype=np.uint64).reshape(nv_total, D, 4)
    if all_have:
        after = np.asarray(step(), dtype=np.uint64).reshape(nv_total, D, 4)
        torch.cuda.synchronize(dev)
        # every rank's two proofs travel to rank 0 (nv x D x 4 words each) and are compared with the CPU proof there
        both = W.gather(np.stack([timed_proof, after]))
        if rank == 0:
            eq_t = [all(bool(np.array_equal(b[0][i], cpu_proof[i])) for b in both) for i in range(nv_total)]
            eq_a = [all(bool(np.array_equal(b[1][i], cpu_proof[i])) for b in both) for i in range(nv_total)]
            parity = {"vs": f"cpu_baseline proof (oracle/oracle.c, all cores), nv={nv_total}, same seed and products"
                            + (f"; the proofs of all {world} ranks" if world > 1 else ""),
                      "rounds_equal": int(sum(eq_t)), "rounds_equal_after_timed_region": int(sum(eq_a)), "rounds": nv_total,
                      "ranks_compared": len(both), "ok": bool(all(eq_t) and all(eq_a))}
    elif sharded_path:
        try:
            from sumcheck_amd import field
            msgs = [sc.ProverMsg(np.asarray(m, dtype=np.uint64).reshape(-1, 4)) for m in timed_proof]
            sub = sc.MLSumcheck.verify(sc.PolynomialInfo(max(len(s_) for s_ in shapes), nv_total), sc.MLSumcheck.extract_sum(msgs), msgs)
            low = np.ascontiguousarray(sub.point[:nv_local])
            mine = torch.stack([sc.DenseMultilinearExtension(nv_local, t).fix_variables(low).evaluations.reshape(4) for t in tables])  # (U, 4)
            parts = W.gather(mine.cpu().numpy().view(np.uint64))
            vals = [[field.to_int(row) for row in pt] for pt in parts]  # vals[g][u]
            high = [field.to_int(x) for x in sub.point[nv_local:]]
            tab_at_point = []
            for u in range(U):
                acc = 0
                for g in range(world):
                    w = 1
                    for j, pj in enumerate(high):  # eq(point_high, g): variable nv_local + j <-> bit j of the rank
                        w = w * (pj if (g >> j) & 1 else (1 - pj)) % field.P
                    acc = (acc + w * vals[g][u]) % field.P
                tab_at_point.append(acc)
            got = 0
            for kk, sh in enumerate(shapes):
                term = field.to_int(coefs[kk])
                for i in sh:
                    term = term * tab_at_point[i] % field.P
                got = (got + term) % field.P
            ok = got == field.to_int(sub.expected_evaluation)
            parity = {"vs": "the verifier (every round accepted, transcript replayed) and its final oracle query, answered from the sharded tables "
                            "(each rank folds its shard on its GPU; ml_sumcheck/test.rs:71-74) -- no CPU proof of this instance exists in this run",
                      "rounds": nv_total, "verifier_accepts": True, "oracle_query_matches": bool(ok), "ok": bool(ok)}
        except Exception as e:
            # (a check that could not RUN is reported, not turned into a failed bench: only a definite mismatch is)
            parity = {"vs": "the verifier and its final oracle query over the sharded tables", "ok": None, "reason": f"the check did not complete: {type(e).__name__}: {e}"}

    # ---- after the clock: keep the GPU busy long enough for an outside sampler to see this run (--min-gpu-seconds of proofs in all: 10 s by default, two periods of a 5 s activity sampler) ----------
    cooldown = 0
    if args.min_gpu_seconds > 0:
        per = elapsed / max(args.steps, 1)
        cooldown = int(min(20000, max(0.0, math.ceil((args.min_gpu_seconds - elapsed) / max(per, 1e-6)))))
        for _ in range(cooldown):
            step()
        torch.cuda.synchronize(dev)

    e2e = None
    if rank == 0 and not sharded_path and not args.no_end_to_end and nv_local <= 24:
        try:
            e2e = end_to_end(sc, _lib, torch, tables, shapes, coefs, nv_local, dev, timed_proof)
        except Exception as e:  # a report beside the line, never a reason to lose it
            e2e = {"failed": f"{type(e).__name__}: {e}"}

    if box["python"]:
        round_loop = "torch.distributed"
    ms, ln = ms_acc, ln_acc
    rounds_ms_total = rounds_ms_acc
    ev_steps = max(timed_steps, 1)

    if rank == 0:
        ops = field_ops(nv_total, shapes, U)
        value = ops * args.steps / elapsed
        # dominant kernel: the merged big-round launch (every product of the round, one product per block row), launched once per BIG
        # round (more than 2^14 pairs on this GPU; later rounds are latency-bound and run through the small-round kernels).
        # Algorithmic bytes of those launches: round 1 reads the tables once; round i >= 2 reads T_{i-1} and writes T_i.
        dom = int(np.argmax(list(ms)))
        merged = ln[0] > 0 and all(ln[q] == 0 for q in range(1, K))
        u_dom = U if merged else len(set(shapes[dom]))
        kname = "k_round_tree" if merged else f"k_prod_tree<{len(shapes[dom])}>"  # (round 1 runs its own instantiation, k_round1_tree)
        # (which rounds those are is taken from the events themselves: the rounds whose launch recorded one)
        big_idx = [i for i in range(1, nv_local + 1) if rln_acc[i - 1] > 0] or list(range(1, max(nv_local - BIG_ROUND_MIN_PAIRS_LOG2, 0) + 1))
        big_rounds = len(big_idx)
        big_bytes = sum(round_bytes(nv_local, u_dom, i) for i in big_idx)
        launches = int(ln[dom])
        avg_ms = ms[dom] / max(launches, 1)
        bytes_per_launch = big_bytes * ev_steps / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # per big round, from the same events: the per-launch average above hides a spread (round 1 is multiplier-bound, the short rounds latency-bound)
        per_round = []
        for i in range(1, nv_local + 1):
            if rln_acc[i - 1] == 0:
                continue
            r_ms = rms_acc[i - 1] / rln_acc[i - 1]
            gb = round_bytes(nv_local, U, i) / 1e9
            per_round.append({"round": i, "kernel": "k_round1_tree_split" if i == 1 else ("k_round_tree_split<chain>" if i == 2 else "k_round_tree_split"),
                              "ms": r_ms, "algorithmic_GB": gb, "frac": (gb / (r_ms * 1e-3)) / HBM_PEAK_GBPS if r_ms > 0 else None,
                              "samples": int(rln_acc[i - 1])})
        # HBM traffic: measured off-line with rocprofv3 PMC passes (tools/profile.sh + tools/collect_profiles.py), per launch of the same
        # kernels; only reported when that file was taken on THIS tree (content hashes of the kernel sources and of this file)
        traffic, traffic_source = None, None
        shas = source_shas()
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")))
            if not (kname in tj.get("kernel", "") and nv_local == 24 and args.config == 3 and world == 1):
                traffic_source = "not reported: the off-line counter passes cover config 3 at nv=24 on one GPU only"
            elif tj.get("csrc_sha") != shas["csrc_sha"] or tj.get("bench_py_sha") != shas["bench_py_sha"]:
                traffic_source = (f"dropped: profiles/hbm_traffic_latest.json was taken on another tree (csrc_sha {tj.get('csrc_sha')} / bench_py_sha "
                                  f"{tj.get('bench_py_sha')}, git {tj.get('git_head')}); running csrc_sha {shas['csrc_sha']} / bench_py_sha {shas['bench_py_sha']}")
            else:
                traffic = tj["traffic_bytes_per_launch"]
                traffic_source = (f"offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this tree (csrc_sha {shas['csrc_sha']}, bench_py_sha "
                                  f"{shas['bench_py_sha']}, git {tj.get('git_head')}), profiles/hbm_traffic_latest.json (not measured in this run)")
        except Exception as e:
            traffic_source = f"not reported: {type(e).__name__}: {e}"
        # The dominant kernel's duration as `rocprofv3 --kernel-trace --stats` of this command sees it (tools/profile.sh: 25 proofs, the
        # tracked profiles/*_rocprofv3_kernel_stats.csv): sum of TotalDurationNs over the big-round kernels / sum of Calls.  It is what
        # `roofline.frac` is computed from when that summary was taken on THIS tree (content hashes), so that the figure can be recomputed
        # from profiles/ alone; the HIP-event figure of this very run stays beside it (frac_hip_events) and tests/test_host.py holds the two
        # within 3 % of each other on the committed set.
        rocprof, rocprof_source = None, None
        try:
            kj = json.load(open(os.path.join(ROOT, "profiles", "rocprof_kernel_latest.json")))
            if not (kname in kj.get("kernel", "") and nv_local == 24 and args.config == 3 and world == 1 and merged):
                rocprof_source = "not used: the off-line rocprofv3 summary covers config 3 at nv=24 on one GPU only"
            elif kj.get("csrc_sha") != shas["csrc_sha"] or kj.get("bench_py_sha") != shas["bench_py_sha"]:
                rocprof_source = (f"dropped: profiles/rocprof_kernel_latest.json was taken on another tree (csrc_sha {kj.get('csrc_sha')} / bench_py_sha "
                                  f"{kj.get('bench_py_sha')}, git {kj.get('git_head')}); running csrc_sha {shas['csrc_sha']} / bench_py_sha {shas['bench_py_sha']}")
            else:
                rocprof = kj
                rocprof_source = (f"rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end` on this tree (csrc_sha {shas['csrc_sha']}, bench_py_sha "
                                  f"{shas['bench_py_sha']}, git {kj.get('git_head')}): {kj.get('stats_csv')}, {kj['launches']} launches, via profiles/rocprof_kernel_latest.json")
        except Exception as e:
            rocprof_source = f"not used: {type(e).__name__}: {e}"
        achieved_events, avg_ms_events = achieved, avg_ms
        if rocprof:
            avg_ms = rocprof["avg_ns"] * 1e-6
            achieved = (big_bytes / big_rounds) / (avg_ms * 1e-3) / 1e9
        # rounds_ms: event span of the rounds launched with events = the big rounds (late rounds are pipelined and record none)
        big_all_bytes = sum(round_bytes(nv_local, U, i) for i in big_idx)
        big_rounds_gbps = big_all_bytes * ev_steps / (rounds_ms_total * 1e-3) / 1e9 if rounds_ms_total > 0 else 0.0
        ms_step = elapsed / args.steps * 1e3
        whole_gbps = algorithmic_bytes(nv_local, U) * args.steps / elapsed / 1e9
        big_kernels_ms = sum(r["ms"] for r in per_round)
        pred = predict_ms(args.config, nv_total, world, U, comm_kind, (exchange or {}).get("exchange_us"))
        pred["measured_over_predicted"] = ms_step / pred["predicted_ms_per_step"]
        cfg_name = ("BASELINE config 4" if args.config == 4 else "BASELINE config 3") + (f", {scaling} scaling" if world > 1 else "")
        ref_muls = reference_muls(nv_total, shapes, U)
        exe = executed_products(nv_local, shapes, U) * world + (executed_products(k, shapes, U) if k else 0)
        out = {
            "metric": "MLSumcheck prover field-ops/s (BLS12-381 Fr, nv=24)" if args.config == 3 else "MLSumcheck prover field-ops/s (BLS12-381 Fr, nv=28, config 4)",
            "value": value, "unit": "field-ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": float(np.median(step_s)) * 1e3, "ms_per_step_min": min(step_s) * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u256 (BLS12-381 Fr, Montgomery form; integer arithmetic on 9 x 29-bit limbs)", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: MLSumcheck prove, ListOfProducts {shapes} over {U} tables, nv={nv_total}"
                                   f" ({nv_local} per GPU shard), BLS12-381 Fr, tables HBM-resident",
                       "nv": nv_total, "nv_per_gpu": nv_local, "tables": U, "degree": max(len(s) for s in shapes),
                       "field_ops_per_step": ops, "sharding": f"high-bit x{world}" if world > 1 else "none",
                       "launcher": W.launcher, "ranks_seen": ranks_seen, "communicator": comm_kind, "exchange": exchange,
                       "round_loop": round_loop, "round_loop_reason": box["why"], "policy": args.policy,
                       # the scaling model's figure for THIS line, written down before any N > 1 hardware run: the line tests it
                       "end_to_end": e2e,
                       "predicted_ms_per_step": pred["predicted_ms_per_step"], "exchange_assumed_us": pred.get("exchange_assumed_us"), "prediction": pred,
                       "gpu_leg": {"warmup_proofs": args.warmup, "timed_proofs": args.steps, "proofs_after_the_clock": cooldown + (1 if all_have else 0),
                                   "event_sampled_proofs": timed_steps,
                                   "ms_per_event_sampled_proof": float(np.mean([t for i, t in enumerate(step_s) if i % every == 0])) * 1e3,
                                   "ms_per_other_proof": float(np.mean([t for i, t in enumerate(step_s) if i % every != 0] or [float("nan")])) * 1e3,
                                   "gpu_leg_seconds_target": args.min_gpu_seconds,
                                   "note": "the proofs after the clock keep the GPU busy for --min-gpu-seconds in all (an outside activity sampler with a 5 s period sees the run); they are not timed"}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_source, "kernel": (f"k_round1_tree_split (round 1) + k_round_tree_split (rounds 2..{big_rounds}): all products, one launch per big round, one product per block row" if merged
                                    else f"{kname} (product {dom}, big rounds)"),
                         "avg_launch_ms": avg_ms, "launches": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
                         # where `achieved` / `frac` come from: the tracked rocprofv3 summary of this command on this tree when there is one
                         # (recomputable from profiles/), else this run's HIP events; the HIP-event figures of THIS run are always here
                         "frac_source": ("rocprofv3: " + rocprof_source) if rocprof else ("HIP events of this run (" + str(rocprof_source) + ")"),
                         "achieved_hip_events": achieved_events, "frac_hip_events": achieved_events / HBM_PEAK_GBPS, "avg_launch_ms_hip_events": avg_ms_events,
                         "rocprof": ({"avg_launch_ms": rocprof["avg_ns"] * 1e-6, "launches": rocprof["launches"], "proofs": rocprof.get("proofs"),
                                      "avg_launch_ms_warm_proofs_only": rocprof["warm"]["avg_ns"] * 1e-6,
                                      "frac_warm_proofs_only": (big_bytes / big_rounds) / (rocprof["warm"]["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBPS,
                                      "per_proof_kernel_total_ms": rocprof.get("per_proof_kernel_total_ms"),
                                      "stats_csv": rocprof.get("stats_csv")} if rocprof else None),
                         "per_round": per_round,
                         "big_rounds_GBps_incl_finalize": big_rounds_gbps, "big_rounds_ms_per_step": rounds_ms_total / ev_steps,
                         "event_timed_steps": timed_steps, "event_timed_every": every,
                         # THE WHOLE PROOF against the same roof (the headline figure: every round, finalize launch and host turn-around
                         # included; `frac` above is the dominant kernel's launches alone, as the contract defines it)
                         "whole_proof_GBps": whole_gbps,  # per GPU
                         "whole_proof_frac": whole_gbps / HBM_PEAK_GBPS,
                         # what the proof pays beside its big-round kernels, per step: the finalize launches inside the big rounds' event span,
                         # and everything outside it (host turn-around of the big rounds: flag over PCIe, hash, bind constants, launch; the
                         # latency-bound rounds below 2^14 pairs).  ms_per_step = big_round_kernels_ms + fixed_cost_ms.
                         "fixed_cost_ms": max(0.0, ms_step - big_kernels_ms),
                         "fixed_cost": {"big_round_kernels_ms": big_kernels_ms,
                                        "finalize_inside_big_rounds_ms": max(0.0, rounds_ms_total / ev_steps - big_kernels_ms),
                                        "turnaround_and_latency_bound_rounds_ms": max(0.0, ms_step - rounds_ms_total / ev_steps),
                                        "latency_bound_rounds": nv_total - big_rounds,
                                        "bytes_at_whole_proof_rate_ms": algorithmic_bytes(nv_local, U) / (HBM_PEAK_GBPS * 1e9) * 1e3},
                         "per_product_ms_per_step": [m / ev_steps for m in ms],
                         # SURVEY 8d's second ceiling: multiplications against what the chip's multiplier can do.  reference_muls_per_s is
                         # the reference ALGORITHM's count over the measured time (a throughput, not a utilisation: the kernels execute
                         # fewer products -- nodes, product tree, claim identity); executed_products_per_s is what the kernels do execute
                         # (binds weighted 97/153), and frac_executed sets that against the carry-free product's in-kernel ceiling.
                         "multiplier": {"reference_muls_per_s": ref_muls * args.steps / elapsed,
                                        "executed_products_per_s": exe * args.steps / elapsed,
                                        "fe_mul_ceiling_per_s": FE_MUL_CEILING_PER_S * world,
                                        "frac_executed": exe * args.steps / elapsed / (FE_MUL_CEILING_PER_S * world),
                                        "executed_products_per_step": exe, "reference_muls_per_step": ref_muls}},
            "cpu_baseline": cpu,
            "parity": parity,
        }
        result["line"] = out
    W.finish()
    result.setdefault("parity_ok", {})[rank] = parity["ok"]


def print_line(out):
    try:  # RCCL prints its NCCL_DEBUG=VERSION banner through C stdio: push it out first so the JSON line is the last line
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def run_threads(args):
    """N thread ranks in this process (ThreadWorld)"""
    N = args.gpus
    shared = ThreadWorld.Shared(N)
    result, errors = {}, [None] * N

    def body(r):
        try:
            run_rank(args, ThreadWorld(shared, r), result)
        except BaseException as e:  # noqa: BLE001
            import traceback
            errors[r] = f"rank {r}: {type(e).__name__}: {e}\n{traceback.format_exc()}"
            try:
                shared.bar.abort()  # nobody waits for a rank that is gone
            except Exception:
                pass

    ts = [threading.Thread(target=body, args=(r,), name=f"rank{r}") for r in range(N)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    real = [e for e in errors if e and "BrokenBarrierError" not in e.split("\n")[0]] or [e for e in errors if e]
    if real:
        log("[bench] thread ranks failed:\n" + "\n".join(real))
        return 1
    print_line(result["line"])
    if result["line"]["parity"]["ok"] is False:
        log("[bench] PARITY FAILURE: the GPU proof differs from the CPU oracle's proof of the same instance")
        return 1
    return 0


def last_json_line(text):
    for l in reversed([l for l in text.splitlines() if l.strip()]):
        if l.lstrip().startswith("{"):
            try:
                return l, json.loads(l)
            except Exception:
                continue
    return None, None


def launcher_env():
    """environment and pre-exec hook for the process that starts the ranks: the CPUs this command started with (see CPUS_AT_START), no
    OpenMP binding variables that only this file set (the launcher would bind ITS main thread and its children, the ranks, inherit)"""
    env = dict(os.environ, SC_BENCH_SELF_LAUNCHED="1")
    if "OMP_NUM_THREADS" not in env:  # (torch.distributed.run would set it to 1 for every rank: rank 0's CPU leg wants the cores)
        hw = len(CPUS_AT_START) if CPUS_AT_START else (os.cpu_count() or 1)  # the CPUs this command may run on, not the host's count
        try:
            from oracle import cref
            q = cref.cpu_quota_cores()
        except Exception:
            q = None
        env["OMP_NUM_THREADS"] = str(max(1, min(hw, q) if q else hw))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in OMP_DEFAULTS_SET_HERE:
        env.pop(k, None)

    def full_mask():  # (runs in the child between fork and exec)
        if CPUS_AT_START:
            try:
                os.sched_setaffinity(0, CPUS_AT_START)
            except OSError:
                pass
    return env, full_mask


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the ranks ourselves"""
    N = args.gpus
    one_gpu = os.environ.get("SC_BENCH_ONE_GPU") == "1"
    try:
        import sumcheck_amd as sc
        n_dev = sc.lib().sc_device_count()
    except Exception as e:
        log(f"[bench] cannot load libsumcheck_hip: {e}")
        return 2
    if n_dev < (1 if one_gpu else N):
        log(f"[bench] --gpus {N} needs {N} visible GPUs, this node shows {n_dev} (SC_BENCH_ONE_GPU=1 runs the N-rank plumbing on one GPU: tests)")
        return 2
    if args.launcher in ("auto", "processes"):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={N}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        log(f"[bench] --gpus {N} without a launcher: starting {N} ranks: {' '.join(cmd[1:9])} bench.py ...")
        env, full_mask = launcher_env()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT, preexec_fn=full_mask)
        line, _ = last_json_line(r.stdout)
        for l in r.stdout.splitlines():  # anything else the ranks wrote to stdout goes to stderr: the JSON line stays the last (and only) stdout line
            if l != line and l.strip():
                log(l)
        if r.returncode == 0 and line:
            print(line, flush=True)
            return 0
        if line and r.returncode != 0:  # (a parity failure: the ranks said so; the line is still the record)
            print(line, flush=True)
            return r.returncode
        if args.launcher == "processes":
            log(f"[bench] torch.distributed.run exited with {r.returncode} and no result line")
            return r.returncode or 1
        log(f"[bench] the process launch failed (exit {r.returncode}); falling back to {N} thread ranks over the peer-to-peer communicator")
    return run_threads(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=(3, 4), help="BASELINE config: 3 = the metric's workload (default), 4 = nv=28 sharded")
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="config 3 at N>1: strong = the nv=24 instance split N ways (the metric as worded), weak = nv=24 per GPU")
    ap.add_argument("--nv", type=int, default=0, help="override the GLOBAL number of variables (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip config.end_to_end (host tables in, proof out; measured after the timed region)")
    ap.add_argument("--time-every", type=int, default=8,
                    help="HIP events around the dominant kernel's launches (the roofline's live duration) on every N-th timed step (at least 4 steps are sampled); 1 = every step")
    ap.add_argument("--launcher", default="auto", choices=("auto", "processes", "threads"),
                    help="--gpus N > 1 without an external launcher: one process per GPU via torch.distributed.run (RCCL), or N thread ranks of this "
                         "process over the library's peer-to-peer communicator; auto = processes, threads if that cannot start")
    ap.add_argument("--policy", action="append", default=[], metavar="KEY=VALUE",
                    help="sc_set_policy(KEY, VALUE) on every rank before anything else (include/sumcheck_hip.h lists the keys), e.g. --policy rccl_direct=0; repeatable")
    ap.add_argument("--min-gpu-seconds", type=float, default=10.0,
                    help="after the timed region, keep proving (untimed) until the GPU leg has lasted about this long; 0 = off")
    args = ap.parse_args()
    if args.gpus < 1 or args.gpus & (args.gpus - 1):
        ap.error("--gpus must be a power of two")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): run `python bench.py --gpus N` (it starts its own ranks) or "
            f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
        sys.exit(2)
    result = {}
    W = ProcWorld()
    run_rank(args, W, result)
    if W.rank == 0:
        print_line(result["line"])
    if any(v is False for v in result.get("parity_ok", {}).values()):
        log("[bench] PARITY FAILURE: the GPU proof differs from the CPU oracle's proof of the same instance")
        sys.exit(1)


if __name__ == "__main__":
    main()
