"""`python bench.py --gpus N` starts its own N ranks: fresh child processes with
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*, started before the parent imports torch; a failed rank ends the
others and the exit code is non-zero.  `--spawn-check` makes every rank report and exit without a GPU."""
import json
import os
import subprocess
import sys
import time

from common import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_n_starts_n_ranks_without_a_launcher():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--spawn-check"], env=_env(), capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert sorted(r["rank"] for r in rows) == [0, 1, 2, 3]
    assert all(r["world"] == 4 and r["local_rank"] == r["rank"] for r in rows)
    assert len({r["master"] for r in rows}) == 1 and rows[0]["master"].startswith("127.0.0.1:")
    assert len({r["pid"] for r in rows}) == 4 and len({r["ppid"] for r in rows}) == 1     # children of one parent
    assert not any(r["torch_imported"] for r in rows)


def test_under_a_launcher_it_is_one_rank():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--spawn-check"],
                         env=_env(RANK="3", LOCAL_RANK="3", WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"),
                         capture_output=True, text=True, timeout=60)
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(rows) == 1 and rows[0]["rank"] == 3 and rows[0]["world"] == 8


def test_a_failed_rank_ends_the_job_with_a_non_zero_exit():
    t0 = time.time()
    out = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--spawn-check"], env=_env(EXA_BENCH_FAIL_RANK="1"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 7
    assert "exited with 7" in out.stderr
    assert time.time() - t0 < 25          # the healthy ranks (sleeping 30 s) were ended, not waited for


def test_pmc_frame_totals_sum_the_kernels_of_a_frame(tmp_path):
    """the live PMC leg of bench.py: per dispatch the counter instances are summed; a frame is one dispatch of the shipped
    march kernel (STATS template argument 0); the wide-march and surfaces pre-pass dispatches are the same frames' work
    and are reported per frame by class; the instrumented variant and other kernels are left out"""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    shipped = "void exa::form1::renderFrameKdKernel<true, true, 0, false, 0, true, 0>(exa::RenderArgs)"
    counted = "void exa::form1::renderFrameKdKernel<true, true, 0, false, 1, false, 0>(exa::RenderArgs)"
    il3 = "void exa::form0::renderFrameKdKernel<true, true, 2, false, 0, true, 3>(exa::RenderArgs)"
    assert all(re.search(bench.FRAME_KERNELS["march"], k) for k in (shipped, il3)) and not re.search(bench.FRAME_KERNELS["march"], counted)
    wide = "void exa::form1::renderFrameKdWideKernel<true, true, false, 4, true>(exa::RenderArgs)"
    pre = "void exa::form1::surfacePrepassKdKernel<0, true, false>(exa::RenderArgs)"
    pre_counted = "void exa::form1::surfacePrepassKdKernel<1, false, false>(exa::RenderArgs)"
    ao = "void exa::form1::aoRaysKdKernel<true>(exa::RenderArgs)"
    assert re.search(bench.FRAME_KERNELS["surfaces_prepass"], pre) and not re.search(bench.FRAME_KERNELS["surfaces_prepass"], pre_counted)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Kernel_Name,Counter_Name,Counter_Value"]
    rows += [f'1,1,0,"{counted}",FETCH_SIZE,1000.0']
    rows += [f'2,2,0,"{shipped}",FETCH_SIZE,10.0', f'2,2,0,"{shipped}",FETCH_SIZE,30.0']       # two instances
    rows += [f'3,3,0,"{shipped}",FETCH_SIZE,60.0', f'3,3,0,"{shipped}",WRITE_SIZE,5.0']
    rows += [f'5,5,0,"{wide}",FETCH_SIZE,8.0', f'6,6,0,"{pre}",FETCH_SIZE,4.0', f'7,7,0,"{pre}",FETCH_SIZE,2.0']
    rows += ['4,4,0,"void exa::untileKernel(unsigned int*)",FETCH_SIZE,7.0']
    f = tmp_path / "1_counter_collection.csv"
    f.write_text("\n".join(rows) + "\n")
    tot, frames = bench.pmc_frame_totals([str(f)])
    assert frames == 2
    assert tot == {"march": {"FETCH_SIZE": 50.0, "WRITE_SIZE": 2.5}, "march_wide": {"FETCH_SIZE": 4.0},
                   "surfaces_prepass": {"FETCH_SIZE": 3.0}}
    # a run bracketed by two marker dispatches: only what lies between them counts, divided by the number of frames the
    # run says it put there — here 2 frames of the split pre-pass plan: two pre-pass and two march dispatches per frame
    # plus one AO launch; warm-up frames in front of the first marker and the counting frame behind the second are left out
    mark = "void exa::profileMarkerKernel(int)"
    rows = [rows[0], f'1,1,0,"{shipped}",SQ_INSTS_VALU,999.0', f'2,2,0,"{mark}",SQ_INSTS_VALU,1.0']
    did = 3
    for frame in range(2):
        for k, v in ((pre, 1.0), (pre, 2.0), (shipped, 10.0), (shipped, 20.0), (ao, 4.0)):
            rows.append(f'{did},{did},0,"{k}",SQ_INSTS_VALU,{v}')
            did += 1
    rows += [f'{did},{did},0,"{mark}",SQ_INSTS_VALU,1.0', f'{did + 1},{did + 1},0,"{counted}",SQ_INSTS_VALU,777.0',
             f'{did + 2},{did + 2},0,"{shipped}",SQ_INSTS_VALU,888.0']
    f.write_text("\n".join(rows) + "\n")
    tot, frames = bench.pmc_frame_totals([str(f)], frames=2)
    assert frames == 2 and tot == {"march": {"SQ_INSTS_VALU": 30.0}, "surfaces_prepass": {"SQ_INSTS_VALU": 3.0},
                                   "ao_rays": {"SQ_INSTS_VALU": 4.0}}
    # the march's kernel name with the walk in it (round 5: <GRAD, FAST, MULTI, SURF, STATS, SMALL, NCH, ROPE>)
    rope = "void exa::form1::renderFrameKdKernel<true, true, 0, false, 0, true, 0, true>(exa::RenderArgs)"
    rope_counted = "void exa::form1::renderFrameKdKernel<true, true, 0, false, 1, false, 0, true>(exa::RenderArgs)"
    assert re.search(bench.FRAME_KERNELS["march"], rope) and not re.search(bench.FRAME_KERNELS["march"], rope_counted)
    lbvh = "void exa::form1::renderFrameKernel<true, true, 0>(exa::RenderArgs)"
    f.write_text(rows[0] + f'\n1,1,0,"{lbvh}",SQ_INSTS_VALU,3.0\n')
    assert bench.pmc_frame_totals([str(f)]) == ({"march": {"SQ_INSTS_VALU": 3.0}}, 1)
    assert bench.pmc_frame_totals([]) == ({}, 0)


def test_pmc_frame_count_of_a_frame_with_surfaces_real_dispatch_list():
    """tests/golden/pmc_c3iso_pass1_counter_collection.csv: the tail of a real `rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU` list
    of a bench.py child on C3 + iso-surface (two timed frames between the two marker dispatches, each frame the split
    pre-pass plan: two pre-pass and two march dispatches; trimmed to the columns the parser reads).  The frame count is the
    caller's (steps x spp): counted from the march dispatches it came out twice too large in round 4."""
    import bench
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pmc_c3iso_pass1_counter_collection.csv")
    tot, frames = bench.pmc_frame_totals([f], frames=2, need_bracket=True)
    assert frames == 2
    assert tot["march"]["SQ_INSTS_VALU"] == 10738573131.0 and tot["surfaces_prepass"]["SQ_INSTS_VALU"] == 265317793.0
    # the fallback (dispatch counting) sees four march dispatches between the markers: half the instructions per "frame"
    tot_fb, frames_fb = bench.pmc_frame_totals([f])
    assert frames_fb == 4 and tot_fb["march"]["SQ_INSTS_VALU"] == tot["march"]["SQ_INSTS_VALU"] / 2
    # ... which is why a bracketed child never falls back: no frame count, or no bracket, is an error
    import pytest
    with pytest.raises(bench.PmcBracketError):
        bench.pmc_frame_totals([f], frames=None, need_bracket=True)
    lines = [ln for ln in open(f).read().splitlines() if "profileMarkerKernel" not in ln]
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix="_counter_collection.csv", delete=False) as g:
        g.write("\n".join(lines) + "\n")
    try:
        with pytest.raises(bench.PmcBracketError):
            bench.pmc_frame_totals([g.name], frames=2, need_bracket=True)
    finally:
        os.unlink(g.name)


def test_pmc_child_that_the_profiler_refuses_is_ended_at_once(tmp_path, monkeypatch):
    """a counter request the hardware cannot serve: the profiler's tool library prints its refusal, aborts the child and then
    hangs in its own finalisation (round 4 record).  live_pmc must not sit out its time limit: it ends the child's process
    group as soon as the refusal shows on stderr and reports it."""
    import shutil
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/bash\necho 'Could not construct profile cfg failed with error code 38: Request exceeds the capabilities "
                    "of the hardware to collect' >&2\nsleep 120\n")
    fake.chmod(0o755)
    monkeypatch.setattr(shutil, "which", lambda name: str(fake))
    t = time.time()
    vals, why = bench.live_pmc(["--steps", "1"], seconds=60.0, frames=1)
    assert vals is None and "refused by the profiler" in why and "error code 38" in why
    assert time.time() - t < 20.0
