"""The boundary is a C ABI: integration/host_loop.c is INTEGRATION.md's host loop as a C99 program that needs nothing but include/prgpu.h
and libprgpu.so.  CPU: it compiles with a plain C compiler, links, and -- there being no HIP device in this container -- fails the way the
product must without a GPU: loudly, with PRGPU_ENODEVICE, never through a CPU path.  GPU: its frame equals the one rendered through the
ctypes mirror, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from pearray_amd import _cabi as abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE = os.path.join(ROOT, "tests", "golden", "scenes", "two_quads.prc")


def _build(tmp_path):
    abi.load()   # (builds the library if it is missing)
    exe = tmp_path / "host_loop"
    lib_dir = os.path.join(ROOT, "pearray_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", "host_loop.c"),
                           "-L", lib_dir, "-lprgpu", "-Wl,-rpath," + lib_dir, "-o", str(exe)])
    return exe


def _has_gpu():
    return abi.load().prgpu_device_count() > 0


def test_c_host_builds_and_fails_loudly_without_a_device(tmp_path):
    exe = _build(tmp_path)
    if _has_gpu():
        pytest.skip("a HIP device is present: the GPU test below runs the program")
    r = subprocess.run([str(exe), SCENE, "2", str(tmp_path / "frame.raw")], capture_output=True, text=True)
    assert r.returncode == 2, (r.returncode, r.stderr)                       # -PRGPU_ENODEVICE
    assert "prgpu_scene_create failed (-2)" in r.stderr and not (tmp_path / "frame.raw").exists()
    r = subprocess.run([str(exe), str(tmp_path / "missing.prc"), "1", str(tmp_path / "frame.raw")], capture_output=True, text=True)
    assert r.returncode != 0 and "prgpu_prc_load_file failed" in r.stderr    # the loader needs no device: its own error comes first


@pytest.mark.gpu
def test_c_host_renders_what_the_ctypes_mirror_renders(tmp_path):
    from pearray_amd import backend, scene
    exe = _build(tmp_path)
    out = tmp_path / "frame.raw"
    r = subprocess.run([str(exe), SCENE, "5", str(out), "96", "64"], capture_output=True, text=True)   # a child process with its own HIP context
    assert r.returncode == 0, r.stderr
    sc = scene.PrcScene(path=SCENE, width=96, height=64)
    g = backend.RenderContext(sc)
    g.render(5)
    g.waitForFinish()
    xyz, smp, _ = g.output()
    raw = np.fromfile(str(out), dtype=np.uint8)
    n = 96 * 64
    assert raw.size == n * 12 + n * 4
    assert np.array_equal(raw[:n * 12].view(np.float32).reshape(64, 96, 3), xyz)
    assert np.array_equal(raw[n * 12:].view(np.uint32).reshape(64, 96), smp)
    st = g.statistics()
    counters = [int(t) for t in r.stdout.split(";")[1].split()]
    assert r.stdout.startswith("96 x 64, 5 iterations;") and len(counters) == 11 and sorted(counters) == sorted(st.values())
    # the PearRay adapter's loop (launches of `lookahead` iterations queued without waiting, a preview fetched after each): the same frame
    ahead = tmp_path / "frame_lookahead.raw"
    r2 = subprocess.run([str(exe), SCENE, "5", str(ahead), "96", "64", "2"], capture_output=True, text=True)
    assert r2.returncode == 0, r2.stderr
    assert "3 launches of <= 2 iterations, 2 previews" in r2.stderr
    assert np.array_equal(np.fromfile(str(ahead), dtype=np.uint8), raw) and r2.stdout == r.stdout
