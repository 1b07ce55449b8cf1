"""Generates tests/golden/cbox_mitsuba_16x16.json from the one reference IMAGE the reference tree ships:
examples/evaluation/cbox.exr, the evaluation scene (examples/evaluation/scene.prc) rendered with Mitsuba 2 (see its README.md).
The fixture is data derived from that data file: linear-sRGB means over a 16 x 16 grid of 16 x 16-pixel blocks.

    python tools/make_cbox_fixture.py [/root/reference]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exr_piz  # noqa: E402

ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
im = exr_piz.read_exr(os.path.join(ref_root, "examples", "evaluation", "cbox.exr"))
rgb = np.stack([im["R"], im["G"], im["B"]], axis=-1)
assert rgb.shape == (256, 256, 3)
blocks = rgb.reshape(16, 16, 16, 16, 3).mean(axis=(1, 3))
out = {"source": "examples/evaluation/cbox.exr (Mitsuba 2 render of examples/evaluation/scene.prc), 256x256 linear RGB, PIZ compressed",
       "layout": "blocks[row][col] = [R, G, B] mean over 16x16 pixels, row 0 = top",
       "blocks": [[[round(float(v), 6) for v in px] for px in row] for row in blocks]}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cbox_mitsuba_16x16.json")
with open(dst, "w") as f:
    json.dump(out, f)
print("wrote", dst)
