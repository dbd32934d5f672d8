"""A fixed set of the seeded random cases of tests/fuzz_cases.py (the sweep tests/gpu_fuzz.py checks any range)."""
import pytest

from gpu_fuzz import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(24))
def test_random_case_matches_oracle(seed):
    bad, desc = check(seed)
    assert not bad, (desc, bad)
