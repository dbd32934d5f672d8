"""The C++ host layer (owlexabrick_amd/host): loaders for the reference's file formats and
the exa::Renderer facade, driven through the headless exaRender CLI."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from common import Case, ROOT, compare
from owlexabrick_amd import harness, scenes

EXE = os.path.join(ROOT, "owlexabrick_amd", "host", "exaRender")


def _run(args):
    return subprocess.run([EXE] + args, capture_output=True, text=True, timeout=300)


def test_config_loaders_and_scalarfield_size_quirk():
    sc = scenes.example("ex3")
    with tempfile.TemporaryDirectory() as d:
        cfg = scenes.write_exa(sc, d, "ex3")
        r = _run([cfg, "--info"])
        assert r.returncode == 0, r.stderr
        out = r.stdout
        assert "bricks 4 cells 200 fields 1" in out
        # ScalarField::load sizes by bytes (4x elements) and folds the zero tail into the range
        assert "elements 800" in out
        lo, hi = [float(x) for x in out.split("range")[1].split()[:2]]
        assert lo == 0.0 and hi == 1.0
        assert "bounds 0 0 0  8 8 4" in out


def test_config_expression_vector_and_errors():
    sc = scenes.example("ex3")
    with tempfile.TemporaryDirectory() as d:
        cfg = scenes.write_exa(sc, d, "ex3")
        with open(cfg, "a") as f:
            f.write('scalar twice expr "%0 2 *"\nvalue_range 0 3\nvector mag ex3_0.scalars ex3_0.scalars ex3_0.scalars # comment\n')
        r = _run([cfg, "--info"])
        assert r.returncode == 0, r.stderr
        assert "fields 3" in r.stdout and "field twice range 0 3" in r.stdout
        assert "field mag range 0 1.73205" in r.stdout
        with open(cfg, "a") as f:
            f.write("bogus_token 1\n")
        r = _run([cfg, "--info"])
        assert r.returncode == 1 and "unknown token 'bogus_token'" in r.stderr
        r = _run([os.path.join(d, "missing.exa"), "--info"])
        assert r.returncode == 1 and "error in opening config file" in r.stderr


def test_exarender_flag_errors_are_reported_before_anything_runs():
    """malformed --devices lists (the loop that never advanced), too many contour planes, unknown flags"""
    for bad in (["--devices", "a"], ["--devices", "0;1"], ["--devices", "0,"], ["--devices", "-1"], ["--devices", ""]):
        r = _run(["x.exa"] + bad)
        assert r.returncode == 1 and "--devices wants a comma-separated list" in r.stderr, (bad, r.stderr)
    r = _run(["x.exa", "--bogus"])
    assert r.returncode == 1 and "unknown flag --bogus" in r.stderr
    r = _run(["x.exa", "--contourplane", "1", "0", "0"])
    assert r.returncode == 1 and "missing value" in r.stderr
    for bad in (["--option", "walk"], ["--option", "=2"], ["--option", "walk=two"], ["--option"]):
        r = _run(["x.exa"] + bad)
        assert r.returncode == 1 and ("--option wants key=integer" in r.stderr or "missing key=value" in r.stderr), (bad, r.stderr)


@pytest.mark.gpu
def test_exarender_option_flag_reaches_the_module():
    """--option key=int = exa_hip_set_option: the stack walk and the rope walk of the region kd-tree give the same file, byte
    for byte, and --stats tells them apart (the rope walk fetches no 16-byte node per leaf); an unknown key is the module's error"""
    sc = scenes.amr(seed=3, root=(2, 2, 2), B=4, levels=3)
    with tempfile.TemporaryDirectory() as d:
        cfg = scenes.write_exa(sc, d, "amr")
        files, nodes = {}, {}
        for walk in (1, 2):
            out = os.path.join(d, f"w{walk}.ppm")
            r = _run([cfg, "--size", "64", "48", "-o", out, "--frames", "2", "--stats", "--option", f"walk={walk}"])
            assert r.returncode == 0, r.stderr
            files[walk] = open(out, "rb").read()
            nodes[walk] = int(r.stdout.split("nodes_visited")[1].split()[0])
        assert files[1] == files[2] and nodes[1] != nodes[2]
        r = _run([cfg, "--size", "64", "48", "--option", "no_such_knob=1"])
        assert r.returncode == 1 and "unknown key no_such_knob" in r.stderr


@pytest.mark.gpu
def test_exarender_contour_plane_flags_match_the_binding():
    """--contourplane nx ny nz offset / --contourchan c (exa/viewer.cpp:1199-1207, the viewer's panel set-up :673-690)"""
    sc = scenes.amr(seed=3, root=(2, 2, 2), B=4, levels=3)
    grey = np.repeat((np.arange(128, dtype=np.float32) / 127.0)[:, None], 4, axis=1)
    W, H = 72, 56
    with tempfile.TemporaryDirectory() as d:
        cfg = scenes.write_exa(sc, d, "amr")
        out = os.path.join(d, "o.ppm")
        r = _run([cfg, "--size", str(W), str(H), "-o", out, "--frames", "1", "--xf-scale", "0.2",
                  "--contourplane", "1", "0.3", "0.2", "0.45", "--contourplane", "0", "0", "1", "0.6"])
        assert r.returncode == 0, r.stderr
        cv = [np.float32(x) for x in r.stdout.split("camera")[1].split()[:12]]
        cam = dict(pos=np.array(cv[0:3]), dir00=np.array(cv[3:6]), dirDu=np.array(cv[6:9]), dirDv=np.array(cv[9:12]))
        data = open(out, "rb").read()
        hdr = f"P6\n{W} {H}\n255\n".encode()
        img = np.frombuffer(data[len(hdr):], dtype=np.uint8).reshape(H, W, 3)[::-1]
    dom = (float(min(sc.fields[0].min(), 0.0)), float(max(sc.fields[0].max(), 0.0)))
    case = Case(sc, W=W, H=H, grad=1, xf=grey, xf_domains=[dom], camera=cam, opacity_scale=0.2,
                contour=[([1, 0.3, 0.2], 0.45, 0), ([0, 0, 1], 0.6, 0)])
    h = case.run_hip()
    assert np.array_equal(harness.unpack_rgba8(h[0])[..., :3], img)
    assert (img.max(axis=-1) > 0).mean() > 0.05            # the planes are in the picture


@pytest.mark.gpu
def test_exarender_matches_binding_and_oracle():
    sc = scenes.amr(seed=3, root=(2, 2, 2), B=4, levels=3)
    grey = np.repeat((np.arange(128, dtype=np.float32) / 127.0)[:, None], 4, axis=1)
    W, H = 80, 48
    with tempfile.TemporaryDirectory() as d:
        tri_v = np.array([[6, 6, 5], [28, 6, 5], [16, 28, 9]], dtype=np.float32)
        tri_t = np.array([[0, 1, 2]], dtype=np.int32)
        cfg = scenes.write_exa(sc, d, "amr", meshes=[(tri_v, tri_t)])
        out = os.path.join(d, "o.ppm")
        r = _run([cfg, "--size", str(W), str(H), "-o", out, "--frames", "1", "--isovals", "0.4", "0.4"])
        assert r.returncode == 0, r.stderr
        assert "Avg. after 1 frames" in r.stdout
        cv = [np.float32(x) for x in r.stdout.split("camera")[1].split()[:12]]
        cam = dict(pos=np.array(cv[0:3]), dir00=np.array(cv[3:6]), dirDu=np.array(cv[6:9]), dirDv=np.array(cv[9:12]))
        data = open(out, "rb").read()
        hdr = f"P6\n{W} {H}\n255\n".encode()
        assert data.startswith(hdr)
        img = np.frombuffer(data[len(hdr):], dtype=np.uint8).reshape(H, W, 3)[::-1]
    dom = (float(min(sc.fields[0].min(), 0.0)), float(max(sc.fields[0].max(), 0.0)))
    case = Case(sc, W=W, H=H, grad=1, xf=grey, xf_domains=[dom], iso=[(0.4, 0), (0.4, 0)], camera=cam,
                meshes=[(tri_v, tri_t)])
    h = case.run_hip()
    assert np.array_equal(harness.unpack_rgba8(h[0])[..., :3], img)      # facade == Python binding, same module
    o = case.run_oracle()
    r = compare(o, h)
    assert r["flips_ok"], r
