"""Streamline tracer: RK4 advection through three scalar fields and rounded-cone
streamline rendering, HIP vs oracle, frame by frame as the viewer drives it (viewer.cpp:281-288)."""
import numpy as np
import pytest

from common import ACCUM_ATOL, Case, compare
from owlexabrick_amd import scenes


def tracer_case(accel):
    sc = scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=3)
    case = Case(sc, W=96, H=80, grad=1, opacity_scale=0.02, accel=accel, fast_math=0)
    rng = np.random.default_rng(2)
    seeds = rng.uniform([8, 8, 8], [24, 24, 24], size=(40, 3)).astype(np.float32)
    return case, seeds


def test_oracle_tracer_moves_points_and_terminates_outside():
    case, seeds = tracer_case(None)
    S = case.oracle_scene()
    S.reset_tracer(True, (0, 1, 2), len(seeds), 6, 6.0, seeds)
    for f in range(5):
        S.advance_tracer()
        fs, P = case.oracle_state(S)
        S.render(fs, P, case.W, case.H, nthreads=4)
    T = S.traces()
    assert np.array_equal(T[:, 0], seeds)
    alive = T[:, 1, 0] < 2e10
    assert alive.any() and np.abs(T[alive, 1] - T[alive, 0]).max() > 0.1     # points moved along the field
    dead = T[..., 0] >= 2e10
    assert (dead[:, 1:] >= dead[:, :-1]).all()                                 # once out, stays out
    assert (T[dead] == 2e10).all()
    assert (T[:, 5] != 0).any()                                                # timesteps 1..5 written (t < NT = 6)


@pytest.mark.gpu
@pytest.mark.parametrize("accel", [1, 0], ids=["kd", "lbvh"])
def test_hip_tracer_matches_oracle_frame_by_frame(accel):
    case, seeds = tracer_case(accel)
    S = case.oracle_scene()
    S.reset_tracer(True, (0, 1, 2), len(seeds), 6, 6.0, seeds)
    R = case.hip_renderer()
    R.resetTracer(seeds, channels=(0, 1, 2), numTimesteps=6, steplen=6.0, enabled=True)
    for f in range(6):
        S.advance_tracer()
        assert R.advanceTracer() == (f + 1 <= 6)
        fs, P = case.oracle_state(S)
        o_rgba, o_acc, o_st = S.render(fs, P, case.W, case.H, nthreads=8)
        R.updateFrameID(0)
        h_rgba = R.render()
        h_acc = R.readAccum()
        assert np.array_equal(R.readTraces(), S.traces()), f                   # RK4 on exact samples: bit-equal
        r = compare((o_rgba, o_acc, o_st), (h_rgba, h_acc, None))
        assert r["accum_bad"] == 0 and r["rgba_bad"] == 0, (f, r)
    assert (np.abs(h_acc[..., :3]).sum() > 0)
    # streamlines stay visible after the tracer is switched off (the BVH persists)
    R.setTracerEnabled(False)
    again = R.render()
    assert np.array_equal(again, h_rgba)
    R.close()
