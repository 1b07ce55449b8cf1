"""CIE tables, spectral nodes and the sRGB->spectrum upsampler against the reference's KATs
(src/core/spectral/CIE.h:21, src/tests/upsampler.cpp:15-106)."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from oracle_binding import f32
from pearray_amd import _cabi as abi
from pearray_amd import scene

WAVELENGTHS = [532, 615, 346, 720, 416]
UPSAMPLER_KATS = [
    ((0.8, 0.2, 0.3), (0.000110479, -0.112288, 27.692141), (0.193251, 0.693976, 0.950077, 0.985873, 0.549449)),
    ((0.2, 0.8, 0.4), (-0.000149, 0.156879, -40.710041), (0.783971, 0.299929, 0.013470, 0.010529, 0.120486)),
    ((0.1, 0.3, 0.8), (0.000033, -0.044228, 13.931887), (0.337471, 0.165104, 0.965322, 0.153193, 0.882958)),
    ((1.0, 1.0, 1.0), (0.0, 0.0, 5e6), (1.0, 1.0, 1.0, 1.0, 1.0)),
    ((0.0, 0.0, 0.0), (0.0, 0.0, -500.0), (0.0, 0.0, 0.0, 0.0, 0.0)),
]


def test_cie_y_norm_sum():
    assert abs(ob.load().orc_cie_y_sum() - 113.042314572337) < 1e-4  # PR_CIE_Y_NORM_SUM, CIE.h:21


def test_cie_eval_shape():
    lib = ob.load()
    xyz = f32(0, 0, 0)
    lib.orc_cie_eval(555.0, xyz)
    assert xyz[1] > xyz[0] > xyz[2] > 0  # photopic peak
    ys = []
    for wl in np.arange(390, 831, 1.0):
        lib.orc_cie_eval(float(wl), xyz)
        ys.append(xyz[1])
    # flat unit spectrum integrates to Y = 1 under the reference normalisation (mean over the 440 nm range)
    assert abs(np.trapezoid(ys, dx=1.0) / 440.0 - 1.0) < 1e-3
    lib.orc_cie_eval(100.0, xyz)  # clamps below the table
    lo = xyz[:]
    lib.orc_cie_eval(390.0, xyz)
    assert lo == xyz[:]


@pytest.mark.parametrize("rgb,coeffs,results", UPSAMPLER_KATS)
def test_upsampler_kats(rgb, coeffs, results):
    """`(refl r g b)` through the product's host routine, evaluated by the oracle (EPS of upsampler.cpp:10)."""
    got = scene.rgb_to_coeffs(rgb)
    assert np.allclose(got, coeffs, atol=1e-4), got
    out = (C.c_float * 5)()
    ob.load().orc_upsample_eval(f32(*got), f32(*WAVELENGTHS), out, 5)
    assert np.allclose(out[:], results, atol=1e-4), out[:]


def test_spectral_nodes():
    b = scene.SceneBuilder(4, 4)
    c = b.spectrum_const(0.25)
    t = b.spectrum_table(400, 700, [0, 8, 15.6, 18.4])
    d = b.illuminant_d65()
    m = b.smul(d, b.spectrum_const(2.0))
    il = b.illum(17, 12, 4)
    mat = b.lambert(c)
    b.add_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]], mat)
    sc = b.build()
    o = ob.OracleScene(sc)
    out = f32(0, 0, 0, 0)
    wl = f32(400, 500, 600, 700)
    o.lib.orc_spectrum_eval(o.h, c, wl, out)
    assert out[:] == [0.25] * 4
    o.lib.orc_spectrum_eval(o.h, t, wl, out)
    assert np.allclose(out[:], [0, 8, 15.6, 18.4], atol=1e-5)
    o.lib.orc_spectrum_eval(o.h, t, f32(350, 450, 750, 900), out)
    assert np.allclose(out[:], [0, 4, 18.4, 18.4], atol=1e-5)  # clamped lookup + lerp
    o.lib.orc_spectrum_eval(o.h, d, f32(560, 560, 300, 830), out)
    assert abs(out[0] - 1.0) < 1e-6  # D65 normalised to 1 at 560 nm
    d560 = out[0]
    o.lib.orc_spectrum_eval(o.h, m, f32(560, 560, 560, 560), out)
    assert out[0] == np.float32(d560) * np.float32(2.0)
    # (illum 17 12 4): scaled fit, power = 2*max (SpectralValueNode.cpp:38-46)
    assert abs(sc.spectra[il].p[3] - 34.0) < 1e-6
    o.lib.orc_spectrum_eval(o.h, il, wl, out)
    assert all(0 <= v <= 34.0 for v in out[:])


def test_coefficient_table_file_matches_the_on_demand_lookup(tmp_path):
    """prgpu_write_rgb_coeff_table writes the SpectralUpsampler format ("SPEC", res, scale, 3 * res^3 * 3 floats,
    SpectralUpsampler.cpp:15-37).  Its grid at resolution 4 is a sub-grid of the resolution-64 table prgpu_rgb_to_coeffs emulates
    (k/3 = 21k/63), so entries must equal the on-demand coefficients of the same colours."""
    import struct
    lib = abi.load()
    path = str(tmp_path / "srgb4.coeff")
    assert lib.prgpu_write_rgb_coeff_table(path.encode(), 4, 4) == 0
    raw = open(path, "rb").read()
    assert raw[:4] == b"SPEC" and struct.unpack_from("<I", raw, 4)[0] == 4 and len(raw) == 8 + 4 * 4 + 3 * 4 ** 3 * 3 * 4
    scale = np.frombuffer(raw, np.float32, 4, 8)
    sm = lambda x: x * x * (3 - 2 * x)
    assert np.allclose(scale, [sm(sm(k / 3)) for k in range(4)], atol=1e-7)
    data = np.frombuffer(raw, np.float32, 3 * 64 * 3, 8 + 16).reshape(3, 4, 4, 4, 3)
    for largest, z, y, x in [(0, 3, 1, 2), (1, 2, 3, 0), (2, 1, 2, 2), (0, 2, 0, 0), (1, 3, 3, 3)]:
        rgb = np.zeros(3, np.float32)
        rgb[largest] = scale[z]
        rgb[(largest + 1) % 3] = np.float32(x / 3) * scale[z]
        rgb[(largest + 2) % 3] = np.float32(y / 3) * scale[z]
        if (rgb >= 1 - 1e-4).all():
            continue                                     # the white special case of prepare() (SpectralUpsampler.cpp:92-97)
        src, dst = (C.c_float * 3)(*rgb), (C.c_float * 3)()
        assert lib.prgpu_rgb_to_coeffs(src, dst) == 0
        wl = np.linspace(400, 700, 7, dtype=np.float32)
        a, b = np.empty(7, np.float32), np.empty(7, np.float32)
        ob.load().orc_upsample_eval(ob.f32(*data[largest, z, y, x]), wl.ctypes.data_as(C.POINTER(C.c_float)), a.ctypes.data_as(C.POINTER(C.c_float)), 7)
        ob.load().orc_upsample_eval(dst, wl.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)), 7)
        assert np.allclose(a, b, atol=2e-3), (largest, z, y, x)
    assert np.allclose(data[0, 0, 1, 1], [0, 0, -50])     # the black row of the table
    assert lib.prgpu_write_rgb_coeff_table(b"/nonexistent-dir/x.coeff", 4, 1) == -5
    assert lib.prgpu_write_rgb_coeff_table(path.encode(), 1, 1) == -1
