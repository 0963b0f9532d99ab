"""GPU parity at BASELINE's full sizes (-m gpu): the headline workload against the CPU oracle, not against itself.

The synthetic tables are generated on the device, copied to the host once and handed to oracle/oracle.c (about 5 s per
nv=24 proof on the GPU box's host cores); every comparison is bit-exact.  Shape of the reference's own end-to-end test
(src/ml_sumcheck/test.rs:64-75: prove, then check every message), with the oracle standing in for "the reference's output".
Also here: a bounded, fixed-seed slice of tools/fuzz.py and tools/fuzz_eval_gkr.py so that the differential runs are part
of the suite."""
import ctypes as C

import numpy as np
import pytest

import sumcheck_amd as sc
from oracle import cref
from sumcheck_amd import _lib
from tests import helpers as H

pytestmark = pytest.mark.gpu

SEED = 0x5C20241008
C3 = (24, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10)       # BASELINE config 3 (the metric's workload), 5 GiB of tables
C3S = (24, [[2, 3, 0, 1], [1, 4, 4], [3, 2, 1], [0, 0]], 5)  # shared tables (shape of reference test.rs:224-252)
C4SHARD = (25, [[0, 1, 2]], 3)                               # one GPU's share of BASELINE config 4 (nv=28 over 8 GPUs)
# products of 8, 7 and 12 multiplicands (the reference's test range, ml_sumcheck/test.rs:125) at a size where one lane of the wide tree kernels
# walks more than 16 / 32 pairs: the running sums' re-canonicalisation inside the grid-stride loop (kernels_wide.hip, kernels_wide16.hip) only runs there
WIDE23 = (23, [[0, 1, 2, 3, 4, 5, 6, 7], [1, 2, 3, 4, 5, 6, 7], [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3]], 8)  # (a product beyond eight: bind pass, canonical tables)
WIDE23F = (23, [[0, 1, 2, 3, 4, 5, 6, 7], [7, 6, 5, 4, 3, 2, 1], [2, 2, 5, 5, 5]], 8)                          # (up to eight: binds fused in, F29 tables, repeats)


def _device_poly(nv, shapes, nt, seed):
    import torch
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    mles = []
    for s in range(nt):
        t = torch.empty((1 << nv, 4), dtype=torch.int64, device=dev)
        _lib.check(sc.lib().sc_synth_table_device(seed, s, 0, 1 << nv, C.c_void_p(t.data_ptr())))
        mles.append(sc.DenseMultilinearExtension(nv, t))
    coefs = cref.synth_table(seed, 1000, len(shapes))
    poly = sc.ListOfProductsOfPolynomials(nv)
    for k, sh in enumerate(shapes):
        poly.add_product([mles[i] for i in sh], coefs[k])
    return poly, mles, coefs


def _oracle_desc(nv, shapes, mles, coefs):
    tabs = [m.evaluations.cpu().numpy().view(np.uint64) for m in mles]
    return H.desc_from(nv, shapes, tabs, coefs)


@pytest.mark.parametrize("nv,shapes,nt", [C3, C3S, C4SHARD, WIDE23, WIDE23F],
                         ids=["config3", "config3_shared", "config4_shard_nv25", "wide_8_7_12_nv23", "wide_8_7_5_nv23"])
def test_full_size_fiat_shamir_proof_bit_exact_vs_oracle(nv, shapes, nt):
    """the whole non-interactive proof (every round polynomial and every challenge) of the production path -- merged big-round
    kernel, F29 bound tables, pipelined late rounds -- equals the oracle's, at the size bench.py measures"""
    poly, mles, coefs = _device_poly(nv, shapes, nt, SEED)
    d = _oracle_desc(nv, shapes, mles, coefs)
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly, borrow=True)
    got = np.stack([m.evaluations for m in proof])
    for i in range(nv):
        assert np.array_equal(got[i], want[i]), f"round {i + 1} of {nv}"
    assert np.array_equal(state.randomness, wrand)
    # a second proof on the rewound handle (what bench.py times) gives the same bits
    state.reset()
    assert np.array_equal(state.prove(sc.Blake2b512Rng.setup()), want)
    # and the borrowed inputs were never written
    for m, t in zip(poly.flattened_ml_extensions, d.tables):  # both in order of first use
        assert np.array_equal(m.evaluations[-4096:].cpu().numpy().view(np.uint64), t[-4096:])


@pytest.mark.parametrize("nv,shapes,nt", [C3, C3S], ids=["config3", "config3_shared"])
def test_full_size_interactive_rounds_and_bound_tables_vs_oracle(nv, shapes, nt):
    """rounds 1..9 through sc_prove_round (the literal prove_round drop-in) against oracle.Prover round by round, with the
    bound tables exported after round 3 (two variables bound: F29 -> canonical through sc_prover_state) and after round 8 (the
    last big round's output, 2^17 entries per table), entry for entry"""
    poly, mles, coefs = _device_poly(nv, shapes, nt, SEED + 1)
    d = _oracle_desc(nv, shapes, mles, coefs)
    chal = cref.synth_table(SEED + 1, 2000, nv)
    op = cref.Prover(d, threads=cref.max_threads())
    st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
    v = None
    for i in range(9):
        want = op.prove_round(None if v is None else v.randomness)
        got = sc.IPForMLSumcheck.prove_round(st, v).evaluations
        assert np.array_equal(got, want), f"round {i + 1}"
        v = sc.VerifierMsg(chal[i])
        if i + 1 in (3, 8):
            _, otabs, _ = op.state()
            gtabs = st.flattened_ml_extensions
            assert len(gtabs) == len(otabs)
            for u, t in enumerate(gtabs):
                assert t.num_vars == nv - i
                assert np.array_equal(t.evaluations, otabs[u]), f"table {u} after round {i + 1}"
    assert st.round == 9
    assert np.array_equal(st.randomness, chal[:8])


C4 = (28, [[0, 1, 2]], 3)  # BASELINE config 4 at its real size: 3 tables x 8 GiB


def _mem_available_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / (1 << 20)
    except Exception:
        pass
    return 0.0


def test_config4_full_size_nv28_unsharded_and_eight_thread_ranks_vs_oracle():
    """BASELINE config 4 -- MLSumcheck prove, one product of three multilinears, nv = 28, 24 GiB of tables -- at its REAL size against
    the oracle (shape of reference src/ml_sumcheck/test.rs:64-75: prove, then check every message):
      (a) sc_ml_prove_sharded with G = 8 thread ranks on GPU 0 over the host transport, every rank borrowing its nv = 25 slice of the
          three device-generated tables (the index arithmetic of the 8-GPU run: slices at > 2^32-byte offsets, 13 sharded rounds, the early
          gather with m = 12, k = 3, 15 replicated rounds) -- whole Fiat-Shamir proof and randomness, identical on every rank;
      (b) the same instance unsharded on one handle (one MI355X holds it);
    both bit for bit equal to cref.ml_prove on the D2H'd tables.  Needs ~60 GiB of host memory for the oracle (tables + its deep copy
    + round 2) and ~50 GiB of HBM."""
    import threading

    import torch
    from sumcheck_amd import sharded
    nv, shapes, nt = C4
    if _mem_available_gib() < 80:
        pytest.skip(f"config 4's oracle run needs ~60 GiB of host memory; MemAvailable = {_mem_available_gib():.0f} GiB")
    sc.lib().sc_release_caches()
    torch.cuda.empty_cache()
    poly, mles, coefs = _device_poly(nv, shapes, nt, SEED)
    d = _oracle_desc(nv, shapes, mles, coefs)
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    last4k = [t[-4096:].copy() for t in d.tables]
    del d  # (the host copies: 24 GiB)
    # (b) one handle, the whole instance
    proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly, borrow=True)
    got = np.stack([m.evaluations for m in proof])
    for i in range(nv):
        assert np.array_equal(got[i], want[i]), f"unsharded: round {i + 1} of {nv}"
    assert np.array_equal(state.randomness, wrand)
    state.close()
    # (a) eight ranks, one thread each, all on GPU 0
    G = 8
    n_loc = (1 << nv) // G
    ex = sharded.ThreadExchange(G)
    out = [None] * G

    def run(rank):
        try:
            _lib.check(sc.lib().sc_set_device(0))
            eng = sharded.HipShardEngine(nv - 3, shapes, coefs, [m.evaluations[rank * n_loc:(rank + 1) * n_loc] for m in mles], "cuda:0", borrow=True)
            comm = ex.comm(rank)
            out[rank] = sharded.prove_sharded_library(eng, comm, nv)
            comm.close()
            eng.close()
        except Exception as e:  # noqa: BLE001
            import traceback
            out[rank] = RuntimeError(f"rank {rank}: {e}\n{traceback.format_exc()}")

    ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=900)
    for r in range(G):
        assert not isinstance(out[r], Exception) and out[r] is not None, out[r]
        sp, sr = out[r]
        for i in range(nv):
            assert np.array_equal(sp[i], want[i]), f"rank {r}: round {i + 1} of {nv}"
        assert np.array_equal(sr, wrand), f"rank {r}"
    # the borrowed inputs were never written
    for m, t in zip(poly.flattened_ml_extensions, last4k):
        assert np.array_equal(m.evaluations[-4096:].cpu().numpy().view(np.uint64), t)
    del poly, mles
    torch.cuda.empty_cache()


FUZZ_SHAPES = 14


def test_fuzz_slice_whole_proofs_vs_oracle():
    """fixed-seed slice of tools/fuzz.py: random product lists (shared tables, repeated factors, 1..6 multiplicands, up to 14
    products) at nv 18..20, so that several big rounds run (merged and per-product launches); whole proofs vs the oracle"""
    rng = np.random.default_rng(20241008)
    for c in range(FUZZ_SHAPES):
        nv = int(rng.choice([18, 19, 20]))
        nt = int(rng.integers(1, 9))
        K = int(rng.integers(13, 15)) if c % 7 == 6 else int(rng.integers(1, 6))
        maxm = 6 if c % 5 == 4 else 4
        shapes = [[int(x) for x in rng.integers(0, nt, size=int(rng.integers(1, maxm + 1)))] for _ in range(K)]
        tabs = [cref.synth_table(9000 + c, s, 1 << nv) for s in range(nt)]
        coefs = cref.synth_table(9000 + c, 1000, K)
        want, wrand = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=cref.max_threads())
        poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0" if c % 2 else None)
        proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly)
        assert np.array_equal(np.stack([m.evaluations for m in proof]), want), (c, nv, nt, shapes)
        assert np.array_equal(state.randomness, wrand)


def test_fuzz_slice_evaluate_and_gkr_vs_oracle():
    """fixed-seed slice of tools/fuzz_eval_gkr.py: sc_poly_evaluate and sc_gkr_prove on random instances vs the oracle"""
    rng = np.random.default_rng(4242)
    for c in range(24):
        nv = int(rng.integers(1, 19))
        nt = int(rng.integers(1, 40))
        K = int(rng.integers(1, 6))
        shapes = [[int(x) for x in rng.integers(0, nt, size=int(rng.integers(1, 6)))] for _ in range(K)]
        tabs = [cref.synth_table(3000 + c, s, 1 << nv) for s in range(nt)]
        coefs = cref.synth_table(3000 + c, 1000, K)
        point = cref.synth_table(3000 + c, 2000, nv)
        poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0" if c % 2 else None)
        assert np.array_equal(poly.evaluate(point), cref.poly_evaluate(H.desc_from(nv, shapes, tabs, coefs), point)), (c, nv, shapes)
    for c in range(16):
        dim = int(rng.integers(1, 15))
        n = 1 << dim
        nnz = int(rng.integers(1, 2 * n + 1))
        idx = np.unique(rng.integers(0, 1 << (3 * dim), size=nnz, dtype=np.uint64))
        vals, f2, f3, g = (cref.synth_table(5000 + c, 1, idx.shape[0]), cref.synth_table(5000 + c, 2, n), cref.synth_table(5000 + c, 3, n),
                           cref.synth_table(5000 + c, 4, dim))
        f1 = sc.SparseMultilinearExtension(3 * dim, idx, vals)
        pr = sc.GKRRoundSumcheck.prove(sc.Blake2b512Rng.setup(), f1, sc.DenseMultilinearExtension(dim, f2), sc.DenseMultilinearExtension(dim, f3), g)
        want, _ = cref.gkr_prove(idx, vals, dim, f2, f3, g, threads=cref.max_threads())
        assert np.array_equal(np.stack([m.evaluations for m in pr.phase1_sumcheck_msgs]), want[0]), (c, dim)
        assert np.array_equal(np.stack([m.evaluations for m in pr.phase2_sumcheck_msgs]), want[1]), (c, dim)
