#!/usr/bin/env python3
"""Reduce the reference's examples/complex.prc (BASELINE config C5; 3.6 MB of scene text) to the arrays the backend consumes and store
them as tests/golden/scenes/complex_c5.npz (inputs only: geometry, materials, camera, settings, lights; the sky light's 5.8 MB
Hosek-Wilkie table is stored as the parameters it is built from -- sun position, turbidity, ground albedo -- and rebuilt on load).  Run where /root/reference exists; the GPU box only sees the .npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pearray_amd import scene  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/examples/complex.prc"
dst = os.path.join(ROOT, "tests", "golden", "scenes", "complex_c5.npz")
s = scene.PrcScene(path=src)
scene.save_scene_npz(dst, s.desc, sky_params=s.sky_params())
back = scene.ArrayScene(dst)
full = np.ctypeslib.as_array(s.desc.spectral_tables, shape=(s.desc.n_spectral_table_values,))
for i in range(s.desc.n_lights):   # the rebuilt sky table equals the one the loader built
    a, b = s.desc.lights[i], back.desc.lights[i]
    if a.kind == scene.abi.LIGHT_SKY:
        n = a.azimuth_count * a.elevation_count * scene.abi.SKY_BANDS
        assert np.array_equal(full[a.table_offset:a.table_offset + n], back.tables[b.table_offset:b.table_offset + n]), "sky table differs"
print("wrote %s: %d triangles, %d entities, %d materials, %d lights, %.2f MB" % (
    dst, back.desc.n_triangles, back.desc.n_entities, back.desc.n_materials, back.desc.n_lights, os.path.getsize(dst) / 1e6))
