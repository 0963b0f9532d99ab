#!/usr/bin/env python3
"""bench.py -- MLSumcheck prover field-ops/s (BLS12-381 Fr) on N MI355X (BASELINE.json's metric).

A "step" is one complete MLSumcheck::prove over HBM-resident synthetic tables: every round's fused bind+sum
kernels, the D2H of the round polynomial, the host Fiat-Shamir hash and the challenge going back as a kernel
argument.  Workload at N=1: BASELINE config 3 (the configuration the metric is quoted on): nv=24, products
[0,1,2,3],[4,5,6],[7,8],[9] over 10 tables (5 GiB).  At N>1 every rank holds a config-3-sized shard
(weak scaling): the global instance has nv = 24 + log2 N variables, tables sharded by the high index bits,
one integer all-reduce of the round polynomial per round over RCCL.

value = field_ops(global instance) / t, field_ops as executed by the reference algorithm:
  (2^nv - 1) * sum_k (2 m_k D + m_k + D) + 3 U (2^nv - 2).

Launch: python bench.py --gpus 1 ...   or   python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x5C20241008
C3_SHAPES = [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def field_ops(nv, shapes, n_tables):
    D = max(len(s) for s in shapes) + 1
    return ((1 << nv) - 1) * sum(2 * len(s) * D + len(s) + D for s in shapes) + 3 * n_tables * ((1 << nv) - 2)


def algorithmic_bytes(nv, n_tables):
    """compulsory HBM traffic of the fused schedule: round 1 reads every table once; round i>=2 reads T_{i-1}
    once and writes T_i once: 32 * U * (4 * 2^nv - 6)"""
    return 32 * n_tables * (4 * (1 << nv) - 6)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(shapes, n_tables, budget_s=15.0):
    """The CPU restatement of the reference algorithm (oracle/oracle.c, rayon-shaped OpenMP) timed on this host's
    cores on a bounded sample of the same workload shape (same products, smaller nv)."""
    from oracle import cref
    threads = cref.max_threads()

    def run(nv):
        tabs = [cref.synth_table(SEED, s, 1 << nv) for s in range(n_tables)]
        coefs = cref.synth_table(SEED, 1000, len(shapes))
        d = cref.PolyDesc(nv, [(coefs[k], s) for k, s in enumerate(shapes)], tabs)
        t0 = time.perf_counter()
        cref.ml_prove(d, threads=threads)
        return time.perf_counter() - t0

    nv = 18
    t = run(nv)
    rate = field_ops(nv, shapes, n_tables) / t
    nv_big = nv
    while nv_big < 24 and field_ops(nv_big + 1, shapes, n_tables) / rate < budget_s:
        nv_big += 1
    if nv_big > nv:
        t = run(nv_big)
        rate = field_ops(nv_big, shapes, n_tables) / t
    while nv_big < 24 and 2.2 * t < budget_s:  # small instances parallelise worse: re-estimate from the last sample
        nv_big += 1
        t = run(nv_big)
        rate = field_ops(nv_big, shapes, n_tables) / t
    return {"value": rate, "unit": "field-ops/s", "cores": threads, "kind": "port",
            "sample": f"same products at nv={nv_big} ({field_ops(nv_big, shapes, n_tables):.3e} field-ops, {t:.1f} s, OpenMP {threads} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nv-local", type=int, default=24, help="variables per GPU shard (24 = BASELINE config 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import sumcheck_amd as sc
    from sumcheck_amd import _lib, sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert world & (world - 1) == 0, "the shard count must be a power of two"
    # SC_BENCH_ONE_GPU=1 (tests only): every rank uses GPU 0 and the ranks exchange through gloo -- the multi-rank plumbing of
    # this file on a one-GPU box (RCCL refuses two ranks on one device, so the rounds run through the torch.distributed loop,
    # the same code the bench falls back to on any RCCL failure).  The numbers of such a run mean nothing.
    one_gpu = os.environ.get("SC_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _lib.check(sc.lib().sc_set_device(local_rank))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    shapes, U = C3_SHAPES, 10
    nv_local = args.nv_local
    k = world.bit_length() - 1
    nv_total = nv_local + k
    n_loc = 1 << nv_local

    # synthetic tables generated on the device; rank g holds entries [g*2^nv_local, (g+1)*2^nv_local) of every table
    tables = []
    for u in range(U):
        t = torch.empty((n_loc, 4), dtype=torch.int64, device=dev)
        _lib.check(sc.lib().sc_synth_table_device(SEED, u, rank * n_loc, n_loc, C.c_void_p(t.data_ptr())))
        tables.append(t)
    ct = torch.empty((len(shapes), 4), dtype=torch.int64, device=dev)
    _lib.check(sc.lib().sc_synth_table_device(SEED, 1000, 0, len(shapes), C.c_void_p(ct.data_ptr())))
    coefs = ct.cpu().numpy().view(np.uint64)
    torch.cuda.synchronize()

    ncomm = None
    state_box = {"ncomm": None}
    force_sharded = os.environ.get("SC_BENCH_FORCE_SHARDED") == "1"  # exercise the N>1 code path on one GPU (tests)
    if world == 1 and not force_sharded:
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
        comm = sharded.DistComm()
        tail_factory = sharded.TailEngines(shapes, coefs, dev)  # the log2(N)-variable tail prover is built once, reloaded per proof

        ncomm = None
        if os.environ.get("SC_BENCH_PYTHON_ROUNDS") != "1" and not one_gpu:
            try:  # per-round all-reduce inside the library (RCCL on the prover's stream); the Python loop is the fallback
                ncomm = sharded.NativeComm(dev)
            except Exception as e:
                log(f"[bench] in-library RCCL rounds unavailable ({e}); using the torch.distributed round loop")
                ncomm = None

        state_box = {"ncomm": ncomm}

        def step():
            engine.reset()
            if state_box["ncomm"] is not None:
                try:
                    return sharded.prove_sharded_native(engine, state_box["ncomm"], comm, nv_total, max(len(s) for s in shapes), tail_factory)[0]
                except Exception as e:  # same on every rank (collective failure): drop to the torch.distributed loop for good
                    log(f"[bench] in-library RCCL rounds failed ({e}); falling back to the torch.distributed round loop")
                    state_box["ncomm"] = None
                    engine.reset()
            return sharded.prove_sharded([engine], comm, nv_total, max(len(s) for s in shapes), tail_factory)[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        proof = step()
    try:  # every rank: flush RCCL's NCCL_DEBUG=VERSION banner (C stdio) now, long before rank 0 prints the JSON line
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    _lib.check(sc.lib().sc_prover_set_timing(handle, 1))  # per-kernel HIP events on the launch stream, timed region only
    barrier()
    t0 = time.perf_counter()
    step_s = []  # a step returns when its last round's message is on the host, so per-step wall times cost nothing extra
    for _ in range(args.steps):
        ts = time.perf_counter()
        proof = step()
        step_s.append(time.perf_counter() - ts)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_gpu else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    K = len(shapes)
    ms = (C.c_double * K)()
    ln = (C.c_uint64 * K)()
    rounds_ms = C.c_double()
    _lib.check(sc.lib().sc_prover_get_timing(handle, ms, ln, C.byref(rounds_ms)))

    if rank == 0:
        ops = field_ops(nv_total, shapes, U)
        value = ops * args.steps / elapsed
        # dominant kernel: k_round_tree (all products; or k_prod_tree<4>, product 0, with SC_MERGE=0).  It is launched once per BIG round (more than 2^16
        # pairs; later rounds are latency-bound and run through k_fix_multi / k_sum_combos).  Algorithmic bytes of those launches:
        # round 1 reads the product's tables once; round i >= 2 reads T_{i-1} and writes T_i (32-byte elements).
        # With every product at <= 4 multiplicands the library runs a big round as ONE launch, k_round_tree, over all products:
        # its event pair is reported under product 0 and the other products report no launches.
        dom = int(np.argmax(list(ms)))
        merged = K > 1 and ln[0] > 0 and all(ln[k] == 0 for k in range(1, K))
        u_dom = U if merged else len(set(shapes[dom]))
        kname = "k_round_tree" if merged else f"k_prod_tree<{len(shapes[dom])}>"
        big_rounds = max(nv_local - 17, 1) if nv_local > 17 else 0
        big_bytes = 32 * u_dom * ((1 << nv_local) + sum((1 << (nv_local - i + 2)) + (1 << (nv_local - i + 1)) for i in range(2, big_rounds + 1)))
        launches = int(ln[dom])
        avg_ms = ms[dom] / max(launches, 1)
        bytes_per_launch = big_bytes * args.steps / max(launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None  # measured off-line with rocprofv3 PMC passes (tools/profile.sh), per launch of the same kernel
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")))
            if tj.get("kernel", "").endswith(kname) and nv_local == 24:
                traffic = tj["traffic_bytes_per_launch"]
        except Exception:
            pass
        # rounds_ms: event span of the rounds launched with events = the big rounds (late rounds are pipelined and record none)
        big_all_bytes = 32 * U * ((1 << nv_local) + sum((1 << (nv_local - i + 2)) + (1 << (nv_local - i + 1)) for i in range(2, big_rounds + 1)))
        big_rounds_gbps = big_all_bytes * args.steps / (rounds_ms.value * 1e-3) / 1e9 if rounds_ms.value > 0 else 0.0
        out = {
            "metric": "MLSumcheck prover field-ops/s (BLS12-381 Fr, nv=24)",
            "value": value, "unit": "field-ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": float(np.median(step_s)) * 1e3, "ms_per_step_min": min(step_s) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 (BLS12-381 Fr, Montgomery form; integer arithmetic on 9 x 29-bit limbs)", "data": "synthetic",
            "config": {"workload": f"MLSumcheck prove, ListOfProducts {shapes} over {U} tables, nv={nv_total}"
                                   f" ({nv_local} per GPU shard), BLS12-381 Fr, tables HBM-resident",
                       "nv": nv_total, "nv_per_gpu": nv_local, "tables": U, "degree": max(len(s) for s in shapes),
                       "field_ops_per_step": ops, "sharding": f"high-bit x{world}" if world > 1 else "none",
                       "round_loop": ("library+rccl" if (world > 1 or force_sharded) and state_box["ncomm"] is not None else
                                      ("torch.distributed" if (world > 1 or force_sharded) else "library"))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "kernel": f"{kname} ({'all products, one launch per big round' if merged else f'product {dom}, big rounds'})",
                         "avg_launch_ms": avg_ms, "launches": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "big_rounds_GBps_incl_finalize": big_rounds_gbps, "big_rounds_ms_per_step": rounds_ms.value / args.steps,
                         "per_product_ms_per_step": [m / args.steps for m in ms],
                         # reference-algorithm multiplications per second over the measured Montgomery-product
                         # ceiling of the chip (137.6 G/s, saturated Comba product, profiles/r1_modmul_ceiling.txt).  It can
                         # exceed 1: the kernels execute fewer products than the reference algorithm (nodes, product tree).
                         "modmul_fraction": (((1 << nv_total) - 1) * sum(len(sh) * (max(len(x) for x in shapes) + 1) for sh in shapes)
                                             + U * ((1 << nv_total) - 2)) * args.steps / elapsed / 137.6e9 / world},
        }
        if world == 1 and not args.no_cpu_baseline and not force_sharded:
            try:
                out["cpu_baseline"] = cpu_baseline(shapes, U)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "field-ops/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        else:
            out["cpu_baseline"] = None
        try:  # RCCL prints its NCCL_DEBUG=VERSION banner through C stdio: push it out first so the JSON line is the last line
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
