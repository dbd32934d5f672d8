"""A fixed set of the seeded random cases of tests/fuzz_cases.py (the sweep tests/gpu_fuzz.py checks any range)."""
import pytest

from gpu_fuzz import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(24))
def test_random_case_matches_oracle(seed):
    bad, desc = check(seed)
    assert not bad, (desc, bad)


# 3342: AO rays of finite length under a magnifying voxel-space transform (a region cut by a trace's tmax is re-entered)
@pytest.mark.parametrize("seed", list(range(16)) + [3342])
def test_random_rich_case_matches_oracle(seed):
    """the second family: AO, contour planes, meshes, voxel-space transform, generated scenes, accumulated frames"""
    bad, desc = check(seed, rich=True)
    assert not bad, (desc, bad)


@pytest.mark.parametrize("seed", range(16))
def test_random_grids_case_matches_oracle(seed):
    """the third family: random partitions into bricks of any shape and level, with holes"""
    bad, desc = check(seed, rich="grids")
    assert not bad, (desc, bad)


@pytest.mark.parametrize("seed", range(10))
def test_random_case_is_independent_of_the_launch(seed):
    """launch order, cost feedback, wide march, shards of a random world + untile, multi-device handle: bit-identical"""
    from gpu_fuzz_sched import check as check_sched
    bad, desc = check_sched(seed)
    assert not bad, (desc, bad)


@pytest.mark.parametrize("seed", range(8))
def test_random_sequence_of_state_changes_equals_a_fresh_renderer(seed):
    """six random setter calls on one handle, each followed by frames that equal a fresh renderer's in that state"""
    from gpu_fuzz_state import check as check_state
    bad, desc = check_state(seed)
    assert not bad, (desc, bad)


@pytest.mark.parametrize("seed", range(8))
def test_random_tracer_run_matches_oracle(seed):
    """streamline tracer: random seeds (also outside the grid), timesteps, step length, channel order; one advance more than
    numTimesteps (the device copy of the timestep stops there, OptixRenderer.cpp:476-487)"""
    from gpu_fuzz_tracer import check as check_tracer
    bad, desc = check_tracer(seed)
    assert not bad, (desc, bad)


@pytest.mark.parametrize("seed", range(8))
def test_random_case_with_three_or_four_fields_matches_oracle(seed):
    bad, desc = check(seed, rich="many")
    assert not bad, (desc, bad)


# 803: a ray whose opacity crosses 0.98 within an ulp of powf on two pixels of one small frame (the reason the flip
# allowance of tests/common.py is "at least 2 pixels per frame")
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 803])
def test_random_case_with_odd_tf_domains_matches_oracle(seed):
    """degenerate, reversed, very narrow, far too wide TF domains"""
    bad, desc = check(seed, rich="domains")
    assert not bad, (desc, bad)


@pytest.mark.parametrize("seed", range(8))
def test_random_deep_scene_matches_oracle(seed):
    """generated scenes of 1e5..2e6 cells: deep kd trees (short-stack restarts), packed leaf references, long rays"""
    bad, desc = check(seed, rich="deep")
    assert not bad, (desc, bad)
