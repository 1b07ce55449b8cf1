"""Lambert eval/pdf/sample identities of src/tests/materials.cpp:48-135 (fixture :13-35: N=+z, V from
(1,0,1)/sqrt2, wavelengths 560/540/400/600, front and back side)."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from oracle_binding import f32
from pearray_amd import scene


@pytest.fixture(scope="module")
def lam():
    b = scene.SceneBuilder(4, 4)
    two = b.lambert(b.refl(0.725, 0.71, 0.68), two_sided=True)
    one = b.lambert(b.refl(0.725, 0.71, 0.68), two_sided=False)
    b.add_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]], two)
    sc = b.build()
    return ob.OracleScene(sc), two, one


WL = f32(560.0, 540.0, 400.0, 600.0)


@pytest.mark.parametrize("backside", [False, True])
def test_sample_matches_eval(lam, backside):
    o, two, one = lam
    s = C.c_uint64()
    o.lib.orc_pcg_seed(42, C.byref(s))
    v = np.array([1, 0, 1.0]) / np.sqrt(2)
    if backside:
        v = -v
    for mat in (two, one):
        for _ in range(16):
            u1, u2 = o.lib.orc_pcg_next_float(C.byref(s)), o.lib.orc_pcg_next_float(C.byref(s))
            l, iw, pdf = f32(0, 0, 0), f32(0, 0, 0, 0), f32(0, 0, 0, 0)
            o.lib.orc_lambert_sample(o.h, mat, WL, f32(*v), u1, u2, l, iw, pdf)
            if mat == one and backside:
                assert list(iw) == [0] * 4 and list(pdf) == [0] * 4  # MaterialSampleOutput::Reject
                continue
            w, epdf = f32(0, 0, 0, 0), f32(0, 0, 0, 0)
            o.lib.orc_lambert_eval(o.h, mat, WL, f32(*v), l, w, epdf)
            assert np.allclose(pdf[:], epdf[:], atol=1e-6)                      # sample.PDF == eval.PDF
            assert np.allclose(np.array(iw[:]) * np.array(pdf[:]), w[:], atol=1e-6)  # IntegralWeight*PDF == Weight
            assert np.signbit(np.float32(l[2])) == np.signbit(v[2])                # same hemisphere as V


def test_eval_opposite_hemisphere_is_zero(lam):
    o, two, _ = lam
    w, pdf = f32(0, 0, 0, 0), f32(0, 0, 0, 0)
    o.lib.orc_lambert_eval(o.h, two, WL, f32(0, 0, 1), f32(0, 0.6, -0.8), w, pdf)
    assert list(w) == [0] * 4 and list(pdf) == [0] * 4
    o.lib.orc_lambert_eval(o.h, two, WL, f32(0, 0, 1), f32(0, 0.6, 0.8), w, pdf)
    assert abs(pdf[0] - 0.8 / np.pi) < 1e-6 and 0 < w[0] < 0.8 / np.pi
