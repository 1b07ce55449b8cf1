#!/usr/bin/env python3
"""Reduce the reference's examples/complex.prc (BASELINE config C5; 3.6 MB of scene text) to the arrays the backend consumes and store
them as tests/golden/scenes/complex_c5.npz (inputs only: geometry, materials, camera, settings, lights WITHOUT the sky table, which is
host supplied).  Run where /root/reference exists; the GPU box only sees the .npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pearray_amd import scene  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/examples/complex.prc"
dst = os.path.join(ROOT, "tests", "golden", "scenes", "complex_c5.npz")
table = np.zeros((256, 512, 11), dtype=np.float32)   # placeholder, dropped again by save_scene_npz
s = scene.PrcScene(path=src, skies={"sky": table})
scene.save_scene_npz(dst, s.desc)
back = scene.ArrayScene(dst, sky_tables=[table])
print("wrote %s: %d triangles, %d entities, %d materials, %d lights, %.2f MB" % (
    dst, back.desc.n_triangles, back.desc.n_entities, back.desc.n_materials, back.desc.n_lights, os.path.getsize(dst) / 1e6))
