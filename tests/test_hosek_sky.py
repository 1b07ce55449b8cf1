"""The sky table of (light :type 'sky'): prgpu_sky_table (csrc/host/skysun.cpp) restates SkyModel::SkyModel (src/skysun/skysun/SkyModel.cpp:
15-56) over the spectral Hosek-Wilkie model (src/skysun/skysun/model/ArHosekSkyModel.cpp).  tests/golden/ref_hosek.json holds the outputs
of the reference's OWN model code, compiled where it lies (oracle/ref/ref_hosek_driver.cpp, tools/make_hosek_golden.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from pearray_amd import _cabi as abi
from pearray_amd import scene

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "ref_hosek.json")) as f:
    CASES = json.load(f)["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_sky_table_equals_the_reference_model(case):
    t = scene.hosek_sky_table(case["sun_elevation"], case["sun_azimuth"], case["turbidity"], case["albedo"], case["elevation_count"], case["azimuth_count"])
    ref = np.array(case["table"], dtype=np.float32).reshape(t.shape)
    assert ref.max() > 0.01
    # same doubles in the same order; the float table is equal bit for bit with this image's libm and within an ulp or two of any other
    assert np.allclose(t, ref, rtol=5e-6, atol=0)
    assert np.array_equal(t, ref) or np.mean(t != ref) < 0.01


def test_sky_table_full_resolution_matches_the_small_one_where_they_coincide():
    c = CASES[0]
    big = scene.hosek_sky_table(c["sun_elevation"], c["sun_azimuth"], c["turbidity"], c["albedo"], 256, 512)
    small = np.array(c["table"], dtype=np.float32).reshape(c["elevation_count"], c["azimuth_count"], 11)
    # row y of an 8-row table is row 32 y of a 256-row one, column x of 16 is column 32 x of 512 (same float angles: y / count)
    assert np.allclose(big[::32, ::32], small, rtol=5e-6)
    assert big.shape == (256, 512, 11) and np.isfinite(big).all() and big.min() >= 0.0


def test_sky_table_rejects_what_the_fit_does_not_cover():
    lib = abi.load()
    alb = (C.c_float * 11)(*[0.2] * 11)
    out = np.zeros(4 * 8 * 11, np.float32)
    p = out.ctypes.data_as(C.POINTER(C.c_float))
    assert lib.prgpu_sky_table(0.5, 1.0, 0.5, alb, 8, 4, p) == -1      # turbidity below 1
    assert lib.prgpu_sky_table(0.5, 1.0, 10.5, alb, 8, 4, p) == -1     # ... above 10
    assert lib.prgpu_sky_table(0.5, 1.0, float("nan"), alb, 8, 4, p) == -1
    assert lib.prgpu_sky_table(0.5, 1.0, 3.0, alb, 0, 4, p) == -1
    assert lib.prgpu_sky_table(0.5, 1.0, 3.0, None, 8, 4, p) == -1
    assert lib.prgpu_sky_table(0.5, 1.0, 10.0, alb, 8, 4, p) == 0 and out.max() > 0


SKY_SCENE = """(scene :render_width 8 :render_height 8
  (camera :name 'c' :type 'standard')
  (light :name 'heaven' :type 'sky' :turbidity 4.5 :albedo %s :elevation 0.9 :azimuth 2.0 :elevation_resolution 6 :azimuth_resolution 12)
  (material :name 'm' :type 'diffuse')
  (mesh :name 'q' (attribute :type 'p' [0,0,0],[1,0,0],[0,1,0]) (faces [0,1,2]))
  (entity :name 'e' :type 'mesh' :mesh 'q' :materials 'm'))"""


@pytest.mark.parametrize("albedo,expect", [("0.3", [0.3] * 11), ("(spectrum :start 320 :end 720 0.0 1.0)", [k / 10 for k in range(11)])])
def test_the_loader_builds_the_table_of_a_sky_light_itself(albedo, expect):
    s = scene.PrcScene(source=SKY_SCENE % albedo)   # no host-supplied table
    params = s.sky_params()
    assert list(params) == [0]
    el, az, turb, alb = params[0]
    assert (np.float32(el), np.float32(az), turb) == (np.float32(0.9), np.float32(2.0), 4.5)
    assert np.allclose(alb, expect, atol=1e-6)
    l = s.desc.lights[0]
    assert l.kind == abi.LIGHT_SKY and (l.azimuth_count, l.elevation_count) == (12, 6)
    tables = np.ctypeslib.as_array(s.desc.spectral_tables, shape=(s.desc.n_spectral_table_values,))
    assert np.array_equal(tables[l.table_offset:l.table_offset + 6 * 12 * 11].reshape(6, 12, 11), scene.hosek_sky_table(el, az, turb, alb, 6, 12))


def test_a_host_supplied_table_still_wins_and_bad_turbidity_is_an_error():
    mine = np.full((6, 12, 11), 0.25, np.float32)
    s = scene.PrcScene(source=SKY_SCENE % "0.3", skies={"heaven": mine})
    l = s.desc.lights[0]
    tables = np.ctypeslib.as_array(s.desc.spectral_tables, shape=(s.desc.n_spectral_table_values,))
    assert np.array_equal(tables[l.table_offset:l.table_offset + mine.size], mine.reshape(-1))
    with pytest.raises(abi.PrgpuError, match="turbidities 1"):
        scene.PrcScene(source=(SKY_SCENE % "0.3").replace(":turbidity 4.5", ":turbidity 12"))
