"""The last GPU test of the suite (files run in alphabetical order): every launch plan the library's host side can choose -- sc_plan_stats,
include/sumcheck_hip.h -- was reached by at least one test of THIS session that also computed the oracle's answer (tests/conftest.py keeps
the table).  Run on its own (or with a -k filter) it has nothing to judge and skips."""
import pytest

from tests import conftest as CT

pytestmark = pytest.mark.gpu

# plans that need what a one-GPU box cannot give are listed with the reason; everything else must be covered
NEEDS_MORE_THAN_THIS_BOX = {}


def test_every_launch_plan_was_compared_with_the_oracle():
    from sumcheck_amd import _lib
    if CT.SESSION["gpu_tests"] < 150:
        pytest.skip(f"only {CT.SESSION['gpu_tests']} GPU tests ran in this session: plan coverage is judged over the whole suite")
    names = list(_lib.plan_stats())
    missing = [n for n in names if not CT.PLAN_TESTS.get(n) and n not in NEEDS_MORE_THAN_THIS_BOX]
    assert not missing, f"launch plans no oracle-comparing test reached: {missing} (reached at all: { {n: CT.PLAN_ANY.get(n, 0) for n in missing} })"
    assert len(names) == _lib.lib().sc_plan_count() >= 30
