#!/usr/bin/env python3
"""tests/golden/ref_hosek.json: outputs of oracle/_ref/ref_hosek_driver -- the reference's own ArHosekSkyModel.cpp, compiled where it lies
and driven like SkyModel::SkyModel drives it -- for a handful of (sun position, turbidity, ground albedo) cases.  Run in the build
container (needs oracle/_ref, i.e. `make -C oracle`); the JSON is committed and pins prgpu_sky_table (tests/test_hosek_sky.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402
import numpy as np  # noqa: E402
from pearray_amd import _cabi as abi  # noqa: E402


def sun(hour, year=2020, month=5, day=6, lat=49.235422, lon=6.9965744, tz=2.0):
    el, az = C.c_float(), C.c_float()
    abi.load().prgpu_sun_position(year, month, day, hour, 0, 0.0, lat, lon, tz, C.byref(el), C.byref(az))
    return el.value, az.value


CASES = [  # name, (elevation, azimuth), turbidity, albedo (scalar or 11 values), azimuth count, elevation count
    ("complex.prc: hour 16, turbidity 3, albedo 0.40", sun(16), 3.0, 0.40, 16, 8),
    ("defaults: hour 12, turbidity 3, albedo 0.15", sun(12), 3.0, 0.15, 12, 6),
    ("low sun, fractional turbidity, coloured ground", (0.1, 5.5), 2.5, [0.05 + 0.06 * k for k in range(11)], 10, 5),
    ("turbidity 10, white ground", (1.2, 0.3), 10.0, 1.0, 8, 4),
    ("turbidity 1, black ground", (0.6, 3.7), 1.0, 0.0, 8, 4),
    ("turbidity 9.75, sun near the zenith", (1.5, 2.0), 9.75, 0.5, 8, 4),
]


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_hosek_driver")
    out = []
    for name, (el, az), turb, alb, azc, elc in CASES:
        alb = [alb] * 11 if not isinstance(alb, list) else alb
        f32 = lambda v: float(np.float32(v)).hex()
        args = [exe, f32(el), f32(az), f32(turb)] + [f32(a) for a in alb] + [str(azc), str(elc)]
        rec = json.loads(subprocess.check_output(args))
        rec["name"] = name
        out.append(rec)
    dst = os.path.join(ROOT, "tests", "golden", "ref_hosek.json")
    with open(dst, "w") as f:
        json.dump({"generator": "oracle/ref/ref_hosek_driver.cpp over /root/reference/src/skysun/skysun/model/ArHosekSkyModel.cpp (tools/make_hosek_golden.py)",
                   "cases": out}, f, indent=None, separators=(",", ":"))
    print("wrote", dst, os.path.getsize(dst), "bytes,", len(out), "cases")


if __name__ == "__main__":
    main()
