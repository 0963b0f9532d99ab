"""CPU tests of the product's host side (no GPU, no compute calls): the C-ABI library loads and exports every
symbol include/sumcheck_hip.h declares, and the host-only entry points (transcript, verifier, lane folding)
agree with the oracle / golden vectors.  Compute entry points must fail loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sumcheck_amd as sc
from oracle import cref
from oracle import pyoracle as po
from sumcheck_amd import _lib, field
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sumcheck_hip.h")).read()
    declared = re.findall(r"SC_API\s+[\w\s\*]+?\b(sc_\w+)\s*\(", hdr)
    assert len(declared) >= 30
    L = C.CDLL(_lib.SO_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/sumcheck_hip.h but not exported"
    assert set(declared) == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"
    assert sc.lib().sc_abi_version() == 5


def test_transcript_matches_golden():
    g = H.load("transcript.json")
    r = sc.Blake2b512Rng.setup()
    for op in g["ops"]:
        if op[0] == "feed":
            r.feed(bytes.fromhex(op[1]))
        elif op[0] == "fill":
            assert r.fill_bytes(op[1]).hex() == op[2]
        else:
            assert field.to_int(r.sample_fr()) == H.hx(op[1])
    s = g["structured"]
    r = sc.Blake2b512Rng.setup()
    r.feed(sc.PolynomialInfo(*s["info"]))
    r.feed(sc.ProverMsg(H.mont(s["msg"])))
    assert field.to_int(r.sample_fr()) == H.hx(s["sample"])
    assert r.fill_bytes(64).hex() == s["next64"]


def test_transcript_blake2b_multi_block_known_answers():
    """The C++ transcript of the product library (csrc/transcript.hpp) on the multi-block inputs of RFC 7693's self-test (0, 3,
    128, 129, 255, 1024 bytes) and on ProverMsg-sized ones (168 = 8 + 5 x 32 bytes spans two compression blocks; "abc" does
    not).  tests/test_oracle.py::test_blake2b_rfc7693_selftest_multi_block ties hashlib to the RFC's published grand hash."""
    import hashlib
    from tests.test_oracle import _rfc7693_seq
    for inlen in (0, 3, 128, 129, 168, 232, 255, 1024):
        data = _rfc7693_seq(inlen, inlen)
        r = sc.Blake2b512Rng.setup()
        if inlen <= 128:
            r.feed(data)
        else:  # fed in uneven pieces: the buffered-block logic must not depend on the split
            r.feed(data[:5]); r.feed(data[5:131]); r.feed(data[131:])
        assert r.fill_bytes(64) == hashlib.blake2b(data, digest_size=64).digest()


def test_transcript_determinism_like_reference():
    # shape of reference src/rng.rs:110-169 (feed / F::rand interleavings incl. unaligned 127/777-byte squeezes)
    rng = np.random.default_rng(5)
    msgs = [rng.bytes(128) for _ in range(7)]

    def run(mk, feed, fill, sample):
        r = mk()
        o = []
        feed(r, msgs[0]); o += [sample(r), sample(r)]
        feed(r, msgs[1]); feed(r, msgs[2]); o.append(sample(r))
        feed(r, msgs[3]); o += [sample(r), sample(r)]
        for m in msgs[4:]:
            feed(r, m)
        f1, f2 = sample(r), sample(r)
        assert f1 != f2
        b1 = fill(r, 127); feed(r, b1)
        b2 = fill(r, 128); b3 = fill(r, 777)
        assert b2[:64] != b3[:64]
        o.append(sample(r)); feed(r, b3); o.append(sample(r))
        return o + [f1, f2]

    a = run(sc.Blake2b512Rng.setup, lambda r, m: r.feed(m), lambda r, n: r.fill_bytes(n), lambda r: field.to_int(r.sample_fr()))
    b = run(sc.Blake2b512Rng.setup, lambda r, m: r.feed(m), lambda r, n: r.fill_bytes(n), lambda r: field.to_int(r.sample_fr()))
    c = run(po.Blake2b512Rng, lambda r, m: r.feed_bytes(m), lambda r, n: r.fill_bytes(n), po.sample_fr)
    assert a == b == c


@pytest.mark.parametrize("name", H.ml_cases())
def test_verifier_accepts_golden_proofs(name):
    case = H.load(name)
    proof = [sc.ProverMsg(H.mont(r)) for r in case["fs_proof"]]
    info = sc.PolynomialInfo(max(len(s) for s in case["shapes"]), case["nv"])
    sub = sc.MLSumcheck.verify(info, H.mont([case["sum"]])[0], proof)
    assert field.to_ints(sub.point) == [H.hx(x) for x in case["fs_randomness"]]
    assert field.to_int(sub.expected_evaluation) == H.hx(case["subclaim_expected"])
    assert field.to_int(sc.MLSumcheck.extract_sum(proof)) == H.hx(case["sum"])
    with pytest.raises(sc.SumcheckError, match="Prover message is not consistent with the claim"):
        sc.MLSumcheck.verify(info, field.from_int(H.hx(case["sum"]) + 1), proof)
    # different transcripts fail (reference test.rs:168-186)
    pr, vr = sc.Blake2b512Rng.setup(), sc.Blake2b512Rng.setup()
    pr.feed(b"Test Trivial Works"); vr.feed(b"Test Trivial Fails")
    if case["nv"] > 1:
        with pytest.raises(sc.SumcheckError):
            sc.MLSumcheck.verify_as_subprotocol(vr, info, H.mont([case["sum"]])[0], proof)


def test_verifier_rejects_non_canonical_encodings_and_short_messages():
    """A proof element (or claimed sum) >= p is not a field element: the reference's Fp cannot hold it.  ev0 + p, ev1 + p would
    hash to the honest transcript and, through a raw 256-bit add that drops the carry, pass P(0) + P(1) == claim for a claim
    that is off by R -- every entry point must refuse the encoding before any arithmetic.  Messages of the wrong length are the
    reference's "incorrect number of evaluations" panic (verifier.rs:60-62)."""
    case = H.load("ml_nv3_c1shape.json")
    info = sc.PolynomialInfo(max(len(s) for s in case["shapes"]), case["nv"])
    good = [H.mont(r) for r in case["fs_proof"]]
    claim = H.mont([case["sum"]])[0]
    pl = np.array([(field.P >> (64 * k)) & ((1 << 64) - 1) for k in range(4)], dtype=np.uint64)

    def plus_p(x):
        v = sum(int(x[k]) << (64 * k) for k in range(4)) + field.P
        assert v < 1 << 256
        return np.array([(v >> (64 * k)) & ((1 << 64) - 1) for k in range(4)], dtype=np.uint64)

    sc.MLSumcheck.verify(info, claim, [sc.ProverMsg(m) for m in good])  # the honest proof passes
    for i in range(case["nv"]):
        for j in range(good[i].shape[0]):
            bad = [m.copy() for m in good]
            bad[i][j] = plus_p(bad[i][j])
            with pytest.raises(sc.SumcheckError, match="canonical") as e:
                sc.MLSumcheck.verify(info, claim, [sc.ProverMsg(m) for m in bad])
            assert e.value.code == _lib.SC_ERR_BAD_ARG
    # the attack of the advisor's finding: both first-round values shifted by p, claim off by one (in Montgomery form: by R)
    bad = [m.copy() for m in good]
    bad[0][0], bad[0][1] = plus_p(bad[0][0]), plus_p(bad[0][1])
    for c in (claim, field.from_int(H.hx(case["sum"]) - 1), field.from_int(H.hx(case["sum"]) + 1)):
        with pytest.raises(sc.SumcheckError):
            sc.MLSumcheck.verify(info, c, [sc.ProverMsg(m) for m in bad])
    with pytest.raises(sc.SumcheckError, match="canonical"):
        sc.MLSumcheck.verify(info, plus_p(claim) if int(claim[3]) < (1 << 62) else pl, [sc.ProverMsg(m) for m in good])
    with pytest.raises(sc.SumcheckError, match="canonical"):
        sc.interpolate_uni_poly(np.stack([pl, good[0][1]]), field.from_int(7))
    with pytest.raises(sc.SumcheckError, match="canonical"):
        sc.interpolate_uni_poly(good[0], pl)
    # every message one evaluation short: stacks fine, must not be read with the wrong stride
    short = [sc.ProverMsg(m[:-1].copy()) for m in good]
    with pytest.raises(sc.SumcheckError, match="incorrect number of evaluations"):
        sc.MLSumcheck.verify(info, claim, short)
    # the C entry point refuses a proof buffer of the wrong length outright
    flat = np.ascontiguousarray(np.stack(good))
    point, exp = np.empty((case["nv"], 4), np.uint64), np.empty(4, np.uint64)
    rc = sc.lib().sc_ml_verify(case["nv"], info.max_multiplicands, C.c_void_p(claim.ctypes.data), C.c_void_p(flat.ctypes.data),
                               flat.shape[0] * flat.shape[1] - 1, None, C.c_void_p(point.ctypes.data), C.c_void_p(exp.ctypes.data))
    assert rc == _lib.SC_ERR_BAD_ARG and b"incorrect number of evaluations" in sc.lib().sc_last_error()


def test_interpolate_uni_poly():
    assert field.to_int(sc.interpolate_uni_poly(field.from_ints([0, 1, 4, 9]), field.from_int(3))) == 9  # verifier.rs:327-331
    rng = np.random.default_rng(4)
    for n in (2, 5, 13, 20, 21, 33, 34, 40):  # the reference's three tiers switch at 20 and 33 points
        coef = [int.from_bytes(rng.bytes(32), "little") % po.P for _ in range(n)]
        f = lambda x: sum(c * pow(x, i, po.P) for i, c in enumerate(coef)) % po.P
        ys = [f(i) for i in range(n)]
        x = int.from_bytes(rng.bytes(32), "little") % po.P
        assert field.to_int(sc.interpolate_uni_poly(field.from_ints(ys), field.from_int(x))) == f(x)
        assert field.to_int(sc.interpolate_uni_poly(field.from_ints(ys), field.from_int(n - 1))) == ys[n - 1]


def test_interpolate_uni_poly_known_answers_over_the_rationals():
    """known answers computed with Python fractions over Q (tests/test_oracle.py: no modular arithmetic in the expectation), lengths on
    both sides of the reference's tier switches (verifier.rs:256-322)"""
    from tests.test_oracle import interpolation_known_answers
    for ys, x, want in interpolation_known_answers():
        assert field.to_int(sc.interpolate_uni_poly(field.from_ints(ys), field.from_int(x))) == want


def test_claim_weights_evaluate_a_polynomial_from_its_kernel_nodes():
    """The host half of the claim identity: f(r) = sum_s lam_s(r) f(node_s) over the kernels' nodes 0, 1, inf, -1, 2 -- what
    turns the previous round's node sums into this round's S(0) + S(1).  Checked against big-integer evaluation of random polynomials."""
    rng = np.random.default_rng(11)
    nodes = [0, 1, None, -1, 2]  # None: the leading coefficient
    for M in (1, 2, 3, 4):
        for _ in range(5):
            coef = [int.from_bytes(rng.bytes(32), "little") % po.P for _ in range(M + 1)]
            f = lambda x: sum(c * pow(x, i, po.P) for i, c in enumerate(coef)) % po.P
            vals = [coef[M] if x is None else f(x % po.P) for x in nodes[: M + 1]]
            for r in (0, 1, po.P - 1, int.from_bytes(rng.bytes(32), "little") % po.P):
                rr, out = field.from_int(r), np.empty((M + 1, 4), np.uint64)
                _lib.check(sc.lib().sc_claim_weights(M, C.c_void_p(rr.ctypes.data), C.c_void_p(out.ctypes.data)))
                lam = field.to_ints(out)
                assert sum(l * v for l, v in zip(lam, vals)) % po.P == f(r), (M, r)
    bad = np.full(4, np.uint64(0xFFFFFFFFFFFFFFFF))
    out = np.empty((5, 4), np.uint64)
    assert sc.lib().sc_claim_weights(2, C.c_void_p(bad.ctypes.data), C.c_void_p(out.ctypes.data)) == _lib.SC_ERR_BAD_ARG
    assert sc.lib().sc_claim_weights(5, C.c_void_p(field.from_int(3).ctypes.data), C.c_void_p(out.ctypes.data)) == _lib.SC_ERR_BAD_ARG


def test_wide_reduce_folds_integer_allreduce_lanes():
    rng = np.random.default_rng(6)
    for ranks in (1, 2, 8, 1000):
        elems = [[int.from_bytes(rng.bytes(32), "little") % po.P for _ in range(ranks)] for _ in range(5)]
        wide = np.zeros((5, 8), dtype=np.uint64)
        for e in range(5):
            for v in elems[e]:
                m = v * po.R % po.P  # Montgomery representation, split into 32-bit limbs, lanes summed as integers
                for j in range(8):
                    wide[e, j] += np.uint64((m >> (32 * j)) & 0xFFFFFFFF)
        out = np.empty((5, 4), dtype=np.uint64)
        _lib.check(sc.lib().sc_wide_reduce(C.c_void_p(wide.ctypes.data), 5, C.c_void_p(out.ctypes.data)))
        assert field.to_ints(out) == [sum(e) % po.P for e in elems]


def test_compute_calls_fail_loudly_without_a_device():
    if sc.lib().sc_device_count() > 0:
        pytest.skip("a HIP device is visible")
    case = H.load("ml_nv3_c1shape.json")
    poly, _ = H.hip_poly(case)
    with pytest.raises(sc.SumcheckError) as e:
        sc.IPForMLSumcheck.prover_init(poly)
    assert e.value.code == _lib.SC_ERR_HIP and "no CPU fallback" in e.value.msg
    with pytest.raises(sc.SumcheckError):
        sc.MLSumcheck.prove(poly)
    with pytest.raises(sc.SumcheckError):
        poly.flattened_ml_extensions[0].fix_variables(field.from_ints([5]))


def test_argument_validation_matches_reference_asserts():
    # data_structures.rs:78 (empty product), :81-84 (wrong num_vars), prover.rs:50-52 (constant)
    poly = sc.ListOfProductsOfPolynomials.new(2)
    t2 = sc.DenseMultilinearExtension(2, np.zeros((4, 4), np.uint64))
    t3 = sc.DenseMultilinearExtension(3, np.zeros((8, 4), np.uint64))
    with pytest.raises(AssertionError):
        poly.add_product([], field.ONE)
    with pytest.raises(AssertionError, match="wrong number of variables"):
        poly.add_product([t2, t3], field.ONE)
    p0 = sc.ListOfProductsOfPolynomials.new(0)
    p0.add_product([sc.DenseMultilinearExtension(0, np.zeros((1, 4), np.uint64))], field.ONE)
    with pytest.raises(sc.SumcheckError, match="Attempt to prove a constant"):
        sc.IPForMLSumcheck.prover_init(p0)
    with pytest.raises(sc.SumcheckError, match="Attempt to prove a constant"):
        sc.MLSumcheck.prove(p0)
    # shared references are stored once (reference test.rs:254)
    p = sc.ListOfProductsOfPolynomials.new(2)
    ts = [sc.DenseMultilinearExtension(2, np.zeros((4, 4), np.uint64)) for _ in range(5)]
    for sh in ([2, 3, 0], [1, 4, 4], [3, 2, 1], [0, 0], [4]):
        p.add_product([ts[i] for i in sh], field.ONE)
    assert len(p.flattened_ml_extensions) == 5 and p.max_multiplicands == 3


def test_header_is_a_c_header_and_the_library_links_from_plain_c(tmp_path):
    """include/sumcheck_hip.h compiled as C99 (-pedantic -Werror) in tests/c/abi_smoke.c, linked against libsumcheck_hip.so and run:
    ABI version, the transcript (BLAKE2b-512("abc")), the verifier's refusal of wrong lengths, and -- without a device -- the loud
    SC_ERR_HIP of a compute entry point.  What a cgo / JNI binding does first; no Python, no torch in that process."""
    import subprocess
    exe = str(tmp_path / "abi_smoke")
    so_dir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe,
                           "-L" + so_dir, "-l:" + os.path.basename(_lib.SO_PATH), "-Wl,-rpath," + so_dir])
    env = dict(os.environ)
    # the pure-C process uses the system HIP runtime (or torch's, when that is the only one present)
    import glob
    extra = [p for p in ("/opt/rocm/lib",) + tuple(glob.glob("/usr/local/lib/python3*/dist-packages/torch/lib")) if os.path.isdir(p)]
    env["LD_LIBRARY_PATH"] = ":".join(extra + [env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0 and "ABI-SMOKE-OK" in r.stdout, (r.returncode, r.stdout, r.stderr[-2000:])


def test_bench_refuses_a_multi_gpu_run_it_cannot_place():
    """`bench.py --gpus N` is self-launching; without N visible devices it says so
    and exits 2 before any rank is started (no silent single-GPU or CPU run)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SC_BENCH_ONE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64",
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=root)
    assert r.returncode == 2, (r.stdout, r.stderr)
    assert "needs 64 visible GPUs" in r.stderr
    assert not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 2 and "power of two" in r.stderr


def test_bench_scaling_model_is_consistent_with_the_sharded_schedule():
    """bench.py's predicted_ms_per_step: T1(nv per GPU) + one exchange per sharded round + the gather + log2(N) extra
    replicated rounds.  The round split it assumes is the library's (protocol.hip: sharded_tail_m, gather at 2^14 pairs): sharded and
    replicated rounds add up to nv, N = 1 is the measured one-GPU time itself, the prediction grows with the exchange time, and the
    variant with this run's measured exchange uses exactly that figure."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json\n"
        f"sys.path.insert(0, {root!r}); sys.argv = ['bench.py']\n"
        "import bench\n"
        "out = {}\n"
        "for cfg, nv, U in ((3, 24, 10), (4, 28, 3)):\n"
        "    for w in (1, 2, 4, 8):\n"
        "        for kind in ('rccl', 'p2p'):\n"
        "            out[f'{cfg}/{w}/{kind}'] = bench.predict_ms(cfg, nv, w, U, kind, exchange_us=7.5)\n"
        "print(json.dumps(out))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for cfg, nv in ((3, 24), (4, 28)):
        one = out[f"{cfg}/1/rccl"]
        assert one["predicted_ms_per_step"] == one["t1_ms"] > 0
        for w in (2, 4, 8):
            k = w.bit_length() - 1
            for kind in ("rccl", "p2p"):
                d = out[f"{cfg}/{w}/{kind}"]
                assert d["sharded_rounds"] + d["replicated_rounds"] == nv
                assert d["replicated_rounds"] == 15                      # the gather leaves 2^14 pairs to the first replicated round
                assert d["sharded_rounds"] == nv - k - (15 - k)
                extra = (d["sharded_rounds"] * d["exchange_assumed_us"] + d["gather_assumed_us"] + d["extra_replicated_rounds_us"]) * 1e-3
                assert abs(d["predicted_ms_per_step"] - (d["t1_ms"] + extra)) < 1e-9
                with_meas = d["predicted_ms_per_step_with_measured_exchange"]
                assert abs((d["predicted_ms_per_step"] - with_meas) - d["sharded_rounds"] * (d["exchange_assumed_us"] - 7.5) * 1e-3) < 1e-9
            assert out[f"{cfg}/{w}/p2p"]["exchange_assumed_us"] <= out[f"{cfg}/{w}/rccl"]["exchange_assumed_us"]
        # strong scaling of a fixed job: the shard's own time falls with N
        assert out[f"{cfg}/8/rccl"]["t1_ms"] < out[f"{cfg}/4/rccl"]["t1_ms"] < out[f"{cfg}/2/rccl"]["t1_ms"] < out[f"{cfg}/1/rccl"]["t1_ms"]


def test_bench_hands_its_ranks_the_cpus_it_started_with():
    """bench.py pins the CPU leg's OpenMP threads (OMP_PROC_BIND); the first OpenMP runtime a process loads then binds its main thread
    to one core, and child processes inherit that mask.  `--gpus N` starts its ranks through a launcher that loads such a runtime
    too: without bench.launcher_env() rank 0's sixteen CPU-leg threads shared one core (seen on the GPU box)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(os.sched_getaffinity(0)) < 2:
        pytest.skip("one CPU")
    prog = (
        "import os, sys, subprocess\n"
        f"sys.path.insert(0, {root!r}); sys.argv = ['bench.py']\n"
        "import bench, torch\n"  # (bench sets the binding variables, torch's OpenMP runtime binds this process's main thread)
        "env, pre = bench.launcher_env()\n"
        "inner = 'import torch, subprocess, sys; subprocess.run([sys.executable, \"-c\", \"import os; print(len(os.sched_getaffinity(0)))\"])'\n"
        "print(len(bench.CPUS_AT_START), flush=True)\n"
        "subprocess.run([sys.executable, '-c', inner], env=env, preexec_fn=pre)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_PROC_BIND", "OMP_PLACES")}
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    started_with, rank_sees = [int(x) for x in r.stdout.split()[-2:]]
    assert started_with == len(os.sched_getaffinity(0)) and rank_sees == started_with, r.stdout


def test_roofline_fraction_follows_from_the_tracked_rocprof_summary():
    """VERDICT r5 item 1.  The committed measurement set -- profiles/bench_line_latest.json (the bench line), profiles/rocprof_kernel_latest.json
    (what bench.py read `roofline.frac` from) and the rocprofv3 stats CSV that summary names -- must tell one story: the line's frac is the
    SURVEY 8(d) formula over the CSV (algorithmic bytes of the big rounds / launches / average duration / 8 TB/s), the HIP-event figure of the
    bench run agrees with it within 3 %, and the kernels of a profiled proof fit inside the step time the driver's clock measured."""
    import csv
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.load(open(os.path.join(root, "profiles", "bench_line_latest.json")))
    kj = json.load(open(os.path.join(root, "profiles", "rocprof_kernel_latest.json")))
    rf = line["roofline"]
    assert rf["frac_source"].startswith("rocprofv3"), rf["frac_source"]  # the line was produced WITH the summary of its own tree
    assert kj["csrc_sha"] in rf["frac_source"] and kj["bench_py_sha"] in rf["frac_source"]
    # the summary is the CSV: every launch of the big-round kernels of the profiled command
    rows = [r for r in csv.DictReader(open(os.path.join(root, kj["stats_csv"]))) if "k_round1_tree" in r["Name"] or "k_round_tree" in r["Name"]]
    calls, total = sum(int(r["Calls"]) for r in rows), sum(int(r["TotalDurationNs"]) for r in rows)
    assert calls == kj["launches"] and total == kj["total_ns"] and calls >= 9 * 20
    nv, U = line["config"]["nv"], line["config"]["tables"]
    big = [x["round"] for x in rf["per_round"]]
    bytes_per_launch = sum(bench.round_bytes(nv, U, i) for i in big) / len(big)
    frac = bytes_per_launch / (total / calls * 1e-9) / 1e9 / bench.HBM_PEAK_GBPS
    assert abs(frac - rf["frac"]) < 1e-9 * frac and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["frac"] / rf["frac_hip_events"] - 1) < 0.03, (rf["frac"], rf["frac_hip_events"])
    assert abs(rf["rocprof"]["frac_warm_proofs_only"] / rf["frac_hip_events"] - 1) < 0.03
    # every kernel of a warm profiled proof (big rounds, finalize launches, the LDS-resident rounds) fits inside the measured step
    assert kj["per_proof_kernel_total_ms"] <= line["ms_per_step"] * 1.03, (kj["per_proof_kernel_total_ms"], line["ms_per_step"])
    assert kj["warm"]["proofs"] >= 20 and kj["per_proof_big_round_kernels_ms"] < kj["per_proof_kernel_total_ms"]
    # the whole proof against the roof is the step time and nothing else
    assert abs(rf["whole_proof_frac"] - bench.algorithmic_bytes(nv, U) / (line["ms_per_step"] * 1e-3) / 1e9 / bench.HBM_PEAK_GBPS) < 1e-9
    # the traffic figure comes from counter passes on the same tree as the line
    tj = json.load(open(os.path.join(root, "profiles", "hbm_traffic_latest.json")))
    assert tj["csrc_sha"] == kj["csrc_sha"] and tj["bench_py_sha"] == kj["bench_py_sha"] and rf["traffic"] == tj["traffic_bytes_per_launch"]
