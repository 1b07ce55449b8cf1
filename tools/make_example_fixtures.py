#!/usr/bin/env python3
"""Reduce every reference example the loader accepts (examples/*.prc; other integrators replaced by `direct`) to the arrays the backend
consumes: tests/golden/scenes/examples/<name>.npz (inputs only: geometry, materials, spectra, camera, settings, lights; a sky light's
5.8 MB Hosek-Wilkie table is stored as the parameters it is built from) at a reduced film.  Run where /root/reference exists; the GPU box only sees the .npz files.
tests/test_reference_examples.py renders them on the GPU against the checker."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pearray_amd import scene, _cabi as abi  # noqa: E402

SKIP = {"complex.prc"}   # BASELINE config C5 has its own fixture (tools/make_c5_fixture.py)
FILM = (96, 64, 4)       # width, height, aa samples


def load(path):
    return scene.PrcScene(path=path, force_direct=True, width=FILM[0], height=FILM[1], spp=FILM[2])


if __name__ == "__main__":
    src_dir = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/examples"
    dst_dir = os.path.join(ROOT, "tests", "golden", "scenes", "examples")
    os.makedirs(dst_dir, exist_ok=True)
    for f in sorted(glob.glob(os.path.join(src_dir, "*.prc"))):
        name = os.path.basename(f)
        if name in SKIP:
            continue
        try:
            s = load(f)
        except abi.PrgpuError as e:
            print("skip  %-28s %s" % (name, str(e)[:110]))
            continue
        dst = os.path.join(dst_dir, name[:-4] + ".npz")
        scene.save_scene_npz(dst, s.desc, sky_params=s.sky_params())
        print("wrote %-28s %7d triangles, %d infinite lights, %.2f MB" % (name, s.desc.n_triangles, s.desc.n_lights, os.path.getsize(dst) / 1e6))
