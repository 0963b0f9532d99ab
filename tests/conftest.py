import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- launch-plan coverage ---------------------------------------------------------------------------------------
# The library counts every path its host side can choose (sc_plan_stats: which big-round kernel, table format, finalize, latency-bound
# form, exchange, GKR initialisation).  Around every GPU test the counters are read, and so is a count of the ORACLE's work (calls into
# oracle/cref.py's provers / evaluators, and loads of the golden fixtures, which are oracle outputs): a test that reached a plan AND
# computed the oracle's answer for what it ran is one that compared that plan with the oracle (every such test asserts equality).
# tests/test_zz_plan_coverage.py, the last GPU test, asserts that every plan was reached that way; the plan -> tests table goes to
# gpurun_out/plan_coverage.json (copied to profiles/ with the round's measurement set).
PLAN_TESTS = {}      # plan name -> test ids that reached it and computed the oracle's answer
PLAN_ANY = {}        # plan name -> launches over the whole session (in this process)
SESSION = {"gpu_tests": 0, "oracle_calls": 0}
_ORACLE_WORK = ("ml_prove", "poly_evaluate", "fix_variables", "gkr_prove", "gkr_phase_one", "gkr_phase_two", "sparse_fix_variables", "ml_verify",
                "check_and_generate_subclaim")


def _count_oracle_work():
    from oracle import cref
    from tests import helpers

    def wrap(fn):
        def inner(*a, **k):
            SESSION["oracle_calls"] += 1
            return fn(*a, **k)
        inner.__wrapped__ = fn
        inner.__name__ = getattr(fn, "__name__", "oracle")
        return inner
    for name in _ORACLE_WORK:
        f = getattr(cref, name)
        if not hasattr(f, "__wrapped__"):
            setattr(cref, name, wrap(f))
    if not hasattr(cref.Prover.prove_round, "__wrapped__"):
        cref.Prover.prove_round = wrap(cref.Prover.prove_round)
    if not hasattr(helpers.load, "__wrapped__"):
        helpers.load = wrap(helpers.load)


@pytest.fixture(autouse=True)
def _plan_coverage(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from sumcheck_amd import _lib
    _count_oracle_work()
    before, calls = _lib.plan_stats(), SESSION["oracle_calls"]
    yield
    after = _lib.plan_stats()
    SESSION["gpu_tests"] += 1
    compared = SESSION["oracle_calls"] > calls
    for name, n in after.items():
        if n > before[name]:
            PLAN_ANY[name] = PLAN_ANY.get(name, 0) + n - before[name]
            if compared:
                PLAN_TESTS.setdefault(name, []).append(request.node.nodeid)


def pytest_sessionfinish(session, exitstatus):
    if not SESSION["gpu_tests"]:
        return
    try:
        from sumcheck_amd import _lib
        names = list(_lib.plan_stats())
        out = {"gpu_tests": SESSION["gpu_tests"], "oracle_calls": SESSION["oracle_calls"],
               "plans": {n: {"launches_in_session": PLAN_ANY.get(n, 0), "oracle_compared_tests": len(PLAN_TESTS.get(n, [])), "tests": PLAN_TESTS.get(n, [])[:12]}
                         for n in names},
               "not_reached_under_an_oracle_comparison": [n for n in names if not PLAN_TESTS.get(n)]}
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "plan_coverage.json"), "w") as f:
            json.dump(out, f, indent=1)
    except Exception as e:  # the report is a by-product: never the reason a session fails
        print(f"[plan coverage] not written: {e}", file=sys.stderr)
