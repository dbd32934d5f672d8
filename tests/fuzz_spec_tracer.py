"""The ORACLE's streamline tracer against the numpy restatement test_oracle_kat._tracer_step_from_spec (CPU only):
    python tests/fuzz_spec_tracer.py FIRST LAST
Random three-field scenes, seeds inside / on / outside the grid, step length, channel order; every step of every trace is
redone from the oracle's previous point with the directions taken from the definition of the basis."""
import sys
import time

import numpy as np

from common import Case
from owlexabrick_amd import scenes
from test_oracle_kat import _tracer_step_from_spec


def check(seed):
    rng = np.random.default_rng(0x7AC35000 + seed)
    B = int(rng.choice([2, 4]))
    levels = int(rng.integers(1, 3))
    root = tuple(int(v) for v in rng.integers(1, 3, 3))
    sc = scenes.amr(seed=int(rng.integers(1, 1 << 20)), root=root, B=B, levels=levels, feature=str(rng.choice(["shell", "plume"])), fields=3)
    ext = np.array(root, dtype=np.float64) * B * (1 << (levels - 1))
    case = Case(sc, W=16, H=8, space_skipping=0, xf_domains=[(0.0, 1.0)] * 3)
    n = int(rng.integers(1, 40))
    seeds = (rng.uniform(-0.1, 1.1, size=(n, 3)) * ext).astype(np.float32)
    nt = int(rng.integers(2, 7))
    steplen = float(rng.choice([0.5, 2.0, 6.0]))
    chans = tuple(int(c) for c in rng.permutation(3))
    S = case.oracle_scene()
    S.reset_tracer(True, chans, n, nt, steplen, seeds)
    worst, compared, moved, ended = 0.0, 0, 0, 0
    prev = S.traces()
    for f in range(nt - 1):
        S.advance_tracer()
        fs, P = case.oracle_state(S)
        S.render(fs, P, case.W, case.H, nthreads=2)
        cur = S.traces()
        for i in range(n):
            info = {}
            want = _tracer_step_from_spec(S, chans, steplen, prev[i, f], info=info)
            if info:
                continue
            got = cur[i, f + 1].astype(np.float64)
            compared += 1
            if want[0] >= 2e10 or got[0] >= 2e10:
                ended += 1
                if not (want[0] >= 2e10 and got[0] >= 2e10):
                    return [f"trace {i} step {f + 1}: ends in one ({got}) and not in the other ({want})"], dict(seed=seed)
                continue
            moved += 1
            worst = max(worst, float(np.abs(want - got).max()))
        prev = cur
    tol = 2e-4 * max(1.0, steplen)
    desc = dict(seed=seed, n=n, nt=nt, steplen=steplen, chans=chans, compared=compared, moved=moved, ended=ended, worst=worst, tol=tol)
    return ([f"a step differs by {worst} > {tol}"] if worst > tol else []), desc


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    fails, moved, t0 = 0, 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed)
        moved += desc.get("moved", 0)
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc} {bad}", flush=True)
    print(f"{fails} failed of {last - first + 1} ({moved} RK4 steps compared), {time.time() - t0:.0f}s")
    sys.exit(1 if fails else 0)
