"""GPU parity of the alternative kernel variants: the production path is the product-tree kernel over internal F29 tables; the
saturated Comba kernel, the node-by-node carry-free kernel, the tiled LDS-staged kernel and the F29-off mode are kept as
cross-checks in the -DSC_EXPERIMENTS build (libsumcheck_hip_exp.so, selected by environment there) and must produce the same bits."""
import os

import numpy as np
import pytest

import sumcheck_amd as sc
from oracle import cref
from tests import helpers as H

pytestmark = pytest.mark.gpu


VARIANT_WORKER = r'''
import json, sys
import numpy as np
import sumcheck_amd as sc
from oracle import cref
from sumcheck_amd import _lib
from tests import helpers as H
assert _lib.SO_PATH.endswith("libsumcheck_hip_exp.so")
for nv, nt, shapes in json.loads(sys.argv[1]):
    tabs = [cref.synth_table(4242 + nv, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(4242 + nv, 1000, len(shapes))
    d = H.desc_from(nv, shapes, tabs, coefs)
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs)
    proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly)
    assert np.array_equal(np.stack([m.evaluations for m in proof]), want), (nv, shapes)
    assert np.array_equal(state.randomness, wrand)
    # interactive flow with a mid-proof state export while the tables are in the internal format (round 2)
    st = sc.IPForMLSumcheck.prover_init(poly)
    op = cref.Prover(d, threads=cref.max_threads())
    chal = cref.synth_table(99, 1, nv)
    v = None
    for i in range(3):
        got = sc.IPForMLSumcheck.prove_round(st, v).evaluations
        assert np.array_equal(got, op.prove_round(None if v is None else v.randomness)), (nv, shapes, i)
        v = sc.VerifierMsg(chal[i])
    _, otabs, _ = op.state()
    for u, t in enumerate(st.flattened_ml_extensions):
        assert np.array_equal(t.evaluations, otabs[u])
print("VARIANT-OK")
'''

VARIANT_SHAPES = [
    (19, 10, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]),       # BASELINE config-3 shape, three big rounds
    (18, 5, [[2, 3, 0, 1], [1, 4, 4], [3, 2, 1], [0, 0]]),  # shared tables, repeated factors
    (18, 6, [[0, 1, 2, 3, 4], [5, 5], [2]]),                # five multiplicands: outside the tree, no F29
    (18, 6, [[0, 1, 2, 3, 4], [5, 5, 1], [2, 3, 4, 0]]),    # products of <= 4 and of 5..8 multiplicands in one round
]


@pytest.mark.parametrize("env", [
    {},                                   # the experiments build's default = the production path
    {"SC_KERNEL": "3", "SC_MERGE": "0"},  # one launch per product
    {"SC_PIPELINE": "0"},                 # late rounds launched after their challenge (no persistent tail kernel, no wait kernel)
    {"SC_FUSED_FIN": "1"},                # big rounds: in-kernel two-level finalize instead of the k_finalize launch (a measured negative result)
    {"SC_SPLIT": "0"},                    # big rounds: every product in every block (k_round_tree) instead of one product per block row
    {"SC_FIN_MB": "0"},                   # big rounds: the single-block k_finalize instead of one block per (product, node)
    {"SC_TAIL": "0"},                     # late rounds as pipelined launches behind the wait kernel (what sharded RCCL proofs use)
    {"SC_KERNEL": "3", "SC_F29": "0"},    # product tree, canonical tables
    {"SC_KERNEL": "0", "SC_FE": "1"},     # node-by-node, carry-free arithmetic
    {"SC_KERNEL": "0", "SC_FE": "0"},     # node-by-node, saturated Comba (inline asm)
    {"SC_FE": "0"},                       # saturated arithmetic with the default kernel family: every fused product of a round on ONE arithmetic
    {"SC_KERNEL": "2"},                   # tiled, LDS-staged
    {"SC_SMALL_LOG2": "13", "SC_GRID": "256"},  # other big/small boundary and grid cap
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) or "default")
def test_variant_matches_oracle(env):
    """The cross-check kernels live in the -DSC_EXPERIMENTS build (libsumcheck_hip_exp.so); each knob setting runs in its own
    process (the knobs are read when a prover is created, the library is chosen when it is first loaded)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if not k.startswith("SC_")}
    e.update(env, SC_LIB_VARIANT="exp")
    r = subprocess.run([sys.executable, "-c", VARIANT_WORKER, json.dumps(VARIANT_SHAPES)], capture_output=True, text=True, timeout=900, cwd=root, env=e)
    assert r.returncode == 0 and "VARIANT-OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_production_library_ignores_the_experiment_knobs(monkeypatch):
    """the shipped library has one production path: SC_KERNEL / SC_FE / SC_F29 / SC_MERGE in the environment change nothing"""
    from sumcheck_amd import _lib
    assert _lib.SO_PATH.endswith("libsumcheck_hip.so")
    for k, v in {"SC_KERNEL": "0", "SC_FE": "0", "SC_F29": "0", "SC_MERGE": "0", "SC_GRID": "7"}.items():
        monkeypatch.setenv(k, v)
    nv, nt, shapes = 18, 10, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]]
    tabs = [cref.synth_table(31, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(31, 1000, len(shapes))
    want, _ = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=cref.max_threads())
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs)
    st = sc.IPForMLSumcheck.prover_init(poly)
    st.set_timing(True)
    got = st.prove()
    assert np.array_equal(got, want)
    ms, launches, _ = st.get_timing()
    assert launches[0] > 0 and all(x == 0 for x in launches[1:]), "big rounds must run as the merged k_round_tree launch"


def test_more_products_than_one_launch_takes():
    """13 products (> kMaxRoundProds = 12): the big rounds fall back to one launch per product; tables shared across products."""
    nv, nt = 18, 4
    shapes = [[0, 1], [1, 2], [2, 3], [3, 0], [0], [1], [2], [3], [0, 2], [1, 3], [0, 1, 2], [1, 2, 3], [0, 1, 2, 3]]
    tabs = [cref.synth_table(9090, s, 1 << nv) for s in range(nt)]
    coefs = cref.synth_table(9090, 1000, len(shapes))
    d = H.desc_from(nv, shapes, tabs, coefs)
    want, wrand = cref.ml_prove(d, threads=cref.max_threads())
    poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs)
    proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly)
    assert np.array_equal(np.stack([m.evaluations for m in proof]), want)
    assert np.array_equal(state.randomness, wrand)


def test_wait_kernel_give_up_is_detected():
    """Pipelined late rounds: if the host does not deliver a challenge within the wait kernel's bound, the round behind it runs on
    a stale challenge.  The library must notice (give-up marker) and void the proof instead of returning wrong messages.  A
    subprocess shortens the bound to one poll (sc_set_policy("wait_spins", 1): process-wide)."""
    import subprocess
    import sys
    code = r'''
import numpy as np, sumcheck_amd as sc
from oracle import cref
from sumcheck_amd import _lib
from tests import helpers as H
_lib.set_policy("wait_spins", 1)
nv, shapes = 12, [[0, 1, 2], [1]]
tabs = [cref.synth_table(7, s, 1 << nv) for s in range(3)]
coefs = cref.synth_table(7, 1000, len(shapes))
poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs)
try:
    sc.MLSumcheck.prove(poly)
    print("NO-ERROR")
except sc.SumcheckError as e:
    print("ERROR", e)
'''
    env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERROR" in r.stdout and "proof is void" in r.stdout, r.stdout + r.stderr[-500:]


def test_wait_kernel_give_up_is_retried_when_the_inputs_survive():
    """Same forced give-up, but on a handle that borrows its (device) tables: the inputs are intact, so sc_ml_prove_handle
    proves again from round 0 with synchronous rounds and the caller gets the right proof (SC_HOST_TRACE reports the retry)."""
    import subprocess
    import sys
    code = r'''
import numpy as np, sumcheck_amd as sc
from oracle import cref
from sumcheck_amd import _lib
from tests import helpers as H
_lib.set_policy("wait_spins", 1)
nv, shapes = 13, [[0, 1, 2], [1]]
tabs = [cref.synth_table(8, s, 1 << nv) for s in range(3)]
coefs = cref.synth_table(8, 1000, len(shapes))
want, wrand = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=4)
poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
proof, state = sc.MLSumcheck.prove_as_subprotocol(sc.Blake2b512Rng.setup(), poly, borrow=True)
ok = np.array_equal(np.stack([m.evaluations for m in proof]), want) and np.array_equal(state.randomness, wrand)
state.reset()
ok2 = np.array_equal(np.asarray(state.prove(sc.Blake2b512Rng.setup())).reshape(want.shape), want)
print("RETRIED-OK" if ok and ok2 else "MISMATCH")
'''
    env = dict(os.environ, SC_HOST_TRACE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RETRIED-OK" in r.stdout, r.stdout + r.stderr[-500:]
    assert "proving again without pipelining" in r.stderr, r.stderr[-1500:]


HAMMER_WORKER = r'''
import ctypes as C, sys, threading, time
import numpy as np
import torch
import sumcheck_amd as sc
from oracle import cref
from sumcheck_amd import _lib
from tests import helpers as H
_lib.set_policy("wait_spins", 20000)
nv, shapes, nt = 16, [[0, 1, 2, 3], [4, 5, 6], [7, 8], [9]], 10
tabs = [cref.synth_table(8800, s, 1 << nv) for s in range(nt)]
coefs = cref.synth_table(8800, 1000, len(shapes))
want, _ = cref.ml_prove(H.desc_from(nv, shapes, tabs, coefs), threads=4)
poly, _ = H.hip_poly_from(nv, shapes, tabs, coefs, device="cuda:0")
hip = C.CDLL("libamdhip64.so")  # the runtime torch (and therefore the library) already mapped
stop, calls = threading.Event(), [0]
def hammer():  # foreign HIP traffic on the same device: allocation, a synchronous copy, a (device-synchronising) free, in a tight loop
    host = (C.c_char * (1 << 20))()
    while not stop.is_set():
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), 1 << 20) == 0
        assert hip.hipMemcpy(p, host, 1 << 20, 1) == 0
        assert hip.hipFree(p) == 0
        calls[0] += 1
th = [threading.Thread(target=hammer) for _ in range(2)]
for t in th: t.start()
t0 = time.time()
st = sc.IPForMLSumcheck.prover_init(poly, borrow=True)
n_ok = 0
for i in range(int(sys.argv[1])):           # device-side waits allowed: a stalled host costs a repeated proof, never a wrong one
    st.reset()
    assert np.array_equal(np.asarray(st.prove()).reshape(want.shape), want), i
    n_ok += 1
_lib.check(sc.lib().sc_prover_set_polling(st._h, 0))
for i in range(int(sys.argv[2])):           # the foreign-host setting: no kernel ever waits for the host
    st.reset()
    assert np.array_equal(np.asarray(st.prove()).reshape(want.shape), want), i
    n_ok += 1
chal = cref.synth_table(8801, 1, nv)
op = cref.Prover(H.desc_from(nv, shapes, tabs, coefs), threads=4)
st.reset()
_lib.check(sc.lib().sc_prover_set_polling(st._h, 1))
v = None
for i in range(nv):                         # the interactive protocol (resident kernel, short patience) under the same traffic
    got = sc.IPForMLSumcheck.prove_round(st, v).evaluations
    assert np.array_equal(got, op.prove_round(None if v is None else v.randomness)), i
    v = sc.VerifierMsg(chal[i])
stop.set()
for t in th: t.join()
print("HAMMER-OK", n_ok, "proofs,", calls[0], "foreign malloc/copy/free cycles,", round(time.time() - t0, 1), "s")
'''


def test_foreign_hip_traffic_never_voids_or_corrupts_a_proof():
    """sumcheck_hip.h's interference contract: two threads of the process hammer hipMalloc / hipMemcpy / hipFree on the prover's
    device while it proves 200 times with device-side waits enabled (a borrowing handle: an expired wait is answered by proving
    again inside the call), 50 times with sc_prover_set_polling(p, 0) (nothing ever waits for the host), and once interactively
    (resident kernel).  Slower is fine; void or wrong is not.  The wait bound is shortened (policy "wait_spins") so that a stall costs
    milliseconds, not seconds, of test time."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if not k.startswith("SC_")}
    r = subprocess.run([sys.executable, "-c", HAMMER_WORKER, "200", "50"], capture_output=True, text=True, timeout=900, cwd=root, env=e)
    assert r.returncode == 0 and "HAMMER-OK 250 proofs" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
