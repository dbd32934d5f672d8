"""The oracle against the committed golden fixtures (tests/golden/, made by make_golden.py)."""
import os
import sys

import numpy as np
import pytest

from common import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import GOLDEN_CASES, crop, golden_window, make_case  # noqa: E402


@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"oracle_{name}.npz"))
    case, frames = make_case(name)
    win = golden_window(name)
    rgba, acc, st = case.run_oracle(nthreads=3, frames=frames, window=win)
    rgba, acc = crop(rgba, win), crop(acc, win)
    S = case.oracle_scene()
    assert S.regions().tobytes() == g["regions"].tobytes()
    assert np.array_equal(S.leaflist(), g["leaflist"])
    assert [st[k] for k in sorted(st)] == g["stats"].tolist()
    assert np.abs(acc - g["accum"]).max() <= 1e-6         # libm powf may differ by an ulp across hosts
    d = np.abs(rgba.view(np.uint8).astype(int) - g["rgba"].view(np.uint8).astype(int))
    assert d.max() <= 1
