"""Known-answer tests restated from the reference's own unit tests: src/tests/tangent.cpp:7-76,
sampling.cpp:7-32, distribution.cpp:9-80, bits.cpp:7-42, entity.cpp:27-48."""
import ctypes as C

import numpy as np

import oracle_binding as ob
from oracle_binding import f32

EPS = 1e-5


def _frame(n, norm=0):
    nx, ny = f32(0, 0, 0), f32(0, 0, 0)
    ob.load().orc_frame_duff(f32(*n), nx, ny, norm)
    return np.array(nx[:]), np.array(ny[:])


def _align(n, v):
    out = f32(0, 0, 0)
    ob.load().orc_tangent_align(f32(*n), f32(*v), out)
    return np.array(out[:])


def test_tangent_frames():
    nx, ny = _frame((0, 0, 1))
    assert np.allclose(nx, (1, 0, 0), atol=EPS) and np.allclose(ny, (0, 1, 0), atol=EPS)
    nx, ny = _frame((0, 1, 0))
    assert np.allclose(nx, (1, 0, 0), atol=EPS) and np.allclose(ny, (0, 0, -1), atol=EPS)
    n = np.array((0, 1, 0.0))
    assert abs(nx @ n) < EPS and abs(ny @ n) < EPS and abs(nx @ ny) < EPS


def test_tangent_align():
    assert np.allclose(_align((0, 0, 1), (0, 1, 0)), (0, 1, 0), atol=EPS)
    assert np.allclose(_align((0, 1, 0), (0, 1, 0)), (0, 0, -1), atol=EPS)
    n = np.array((0, 0.5, 0.5)) / np.linalg.norm((0, 0.5, 0.5))
    assert np.allclose(_align(n, (0, 1, 0)), np.array((0, 0.5, -0.5)) / np.linalg.norm((0, 0.5, -0.5)), atol=EPS)
    assert np.allclose(_align((1, 0, 0), (0, 1, 0)), (0, 1, 0), atol=EPS)


def test_tangent_space_roundtrip():
    lib = ob.load()
    rng = np.random.default_rng(1)
    for _ in range(50):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        nx, ny = _frame(n, 1)
        v = rng.normal(size=3); v /= np.linalg.norm(v)
        t, back = f32(0, 0, 0), f32(0, 0, 0)
        lib.orc_to_tangent_space(f32(*n), f32(*nx), f32(*ny), f32(*v), t)
        lib.orc_from_tangent_space(f32(*n), f32(*nx), f32(*ny), t, back)
        assert np.allclose(back[:], v, atol=1e-5)


def test_cos_hemi_normalised():
    out = f32(0, 0, 0)
    ob.load().orc_cos_hemi(0.5, 0.5, out)
    assert abs(np.dot(out[:], out[:]) - 1) < EPS
    rng = np.random.default_rng(2)
    for u1, u2 in rng.random((200, 2)):
        ob.load().orc_cos_hemi(float(u1), float(u2), out)
        assert abs(np.dot(out[:], out[:]) - 1) < 1e-5 and out[2] >= 0
        assert abs(out[2] - np.sqrt(np.float32(u1))) < 1e-6


def test_sincos_2pi_accuracy():
    lib = ob.load()
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for u in np.linspace(0, 1, 4001, endpoint=False, dtype=np.float32):
        lib.orc_sincos_2pi(float(u), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(2 * np.pi * np.float64(u))), abs(c.value - np.cos(2 * np.pi * np.float64(u))))
    assert worst < 3e-7


def _cdf(values):
    v = np.asarray(values, dtype=np.float32)
    cdf = np.zeros(len(v) + 1, dtype=np.float32)
    total = C.c_float()
    ob.load().orc_distribution_generate(v.ctypes.data_as(C.POINTER(C.c_float)), len(v), cdf.ctypes.data_as(C.POINTER(C.c_float)), C.byref(total))
    return cdf, total.value


def test_distribution_kats():
    lib = ob.load()
    cdf, total = _cdf([0, 1, 2, 3, 4])
    assert total == 10.0
    cdf, _ = _cdf([1] * 5)
    assert np.allclose(np.diff(cdf)[[0, 2, 4]], 0.2, atol=EPS)
    cdf, _ = _cdf([i + 1.0 for i in range(5)])
    assert np.allclose(np.diff(cdf)[[0, 2, 4]], [1 / 15, 3 / 15, 5 / 15], atol=EPS)
    cdf, _ = _cdf([1] * 5)
    p = cdf.ctypes.data_as(C.POINTER(C.c_float))
    for x in (0.25, 0.5, 0.75):
        assert abs(lib.orc_distribution_continuous_pdf(p, 6, x) - 1.0) < EPS
    cdf, _ = _cdf([(i / 16.0) ** 2 for i in range(5)])
    p = cdf.ctypes.data_as(C.POINTER(C.c_float))
    pdf, rem = C.c_float(), C.c_float()
    x = lib.orc_distribution_sample_discrete(p, 6, 0.5, C.byref(pdf), C.byref(rem))
    assert pdf.value == np.float32(cdf[x + 1] - cdf[x])
    xc = lib.orc_distribution_sample_continuous(p, 6, 0.5, C.byref(pdf))
    assert lib.orc_distribution_continuous_pdf(p, 6, xc) == pdf.value


def test_distribution_edges():
    lib = ob.load()
    cdf, _ = _cdf([0, 0, 0])  # degenerate -> uniform
    assert np.allclose(cdf, [0, 1 / 3, 2 / 3, 1])
    p = cdf.ctypes.data_as(C.POINTER(C.c_float))
    pdf = C.c_float()
    assert lib.orc_distribution_sample_discrete(p, 4, 0.0, C.byref(pdf), None) == 0
    assert lib.orc_distribution_sample_discrete(p, 4, 0.99999994, C.byref(pdf), None) == 2


def test_morton():
    lib = ob.load()
    x, y = C.c_uint32(), C.c_uint32()
    lib.orc_morton_2_xy(lib.orc_xy_2_morton(42, 56), C.byref(x), C.byref(y))
    assert (x.value, y.value) == (42, 56)
    lib.orc_morton_2_xy(lib.orc_xy_2_morton(1548, 65535), C.byref(x), C.byref(y))
    assert (x.value, y.value) == (1548, 65535)
    assert [lib.orc_xy_2_morton(a, b) for a, b in ((0, 0), (1, 0), (0, 1), (1, 1), (2, 0))] == [0, 1, 2, 3, 4]


def test_normal_matrix_and_nonuniform_scale():
    lib = ob.load()
    out, det = (C.c_float * 9)(), C.c_float()
    lib.orc_normal_matrix(f32(*np.eye(4).ravel()), out, C.byref(det))
    assert list(out) == [1, 0, 0, 0, 1, 0, 0, 0, 1] and det.value == 1.0
    # entity.cpp:38-47: position (0,1,1), rotation 90deg about z, scale (1,2,1): (1,1,1) -> (-2,2,2)
    c, s = np.cos(np.pi / 2), np.sin(np.pi / 2)
    M = np.eye(4); M[:3, :3] = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) @ np.diag([1, 2, 1]); M[:3, 3] = (0, 1, 1)
    assert np.allclose(M @ [1, 1, 1, 1], [-2, 2, 2, 1], atol=1e-6)
    lib.orc_normal_matrix(f32(*M.astype(np.float32).ravel()), out, C.byref(det))
    assert np.allclose(np.array(out[:]).reshape(3, 3), np.linalg.inv(M[:3, :3]).T, atol=1e-6) and abs(det.value - 2) < 1e-6


def test_safe_position_moves_off_surface():
    lib = ob.load()
    out = f32(0, 0, 0)
    lib.orc_safe_position(f32(1, 2, 3), f32(0, 0, 1), f32(0, 0, 1), out)
    assert out[0] == 1 and out[1] == 2 and out[2] > 3 and out[2] - 3 < 2e-4
    lib.orc_safe_position(f32(1, 2, 3), f32(0, 0, -1), f32(0, 0, 1), out)
    assert out[2] < 3


def test_triangle_sample_inside():
    lib = ob.load()
    rng = np.random.default_rng(3)
    out = f32(0, 0)
    for u in rng.random((200, 2)):
        lib.orc_triangle_sample(f32(*u), out)
        assert out[0] >= 0 and out[1] >= 0 and out[0] + out[1] <= 1 + 1e-6


def test_filter_tables():
    lib = ob.load()
    from pearray_amd import _cabi as abi
    for kind in (abi.FILTER_BLOCK, abi.FILTER_TRIANGLE, abi.FILTER_GAUSSIAN, abi.FILTER_MITCHELL):
        for r in (0, 1, 2, 3):
            t = np.zeros((2 * r + 1) ** 2, dtype=np.float32)
            lib.orc_filter_table(kind, r, t.ctypes.data_as(C.POINTER(C.c_float)))
            assert abs(t.sum() - 1.0) < 1e-5, (kind, r, t.sum())
            assert np.allclose(t.reshape(2 * r + 1, -1), t.reshape(2 * r + 1, -1).T)
    # the reference default (Mitchell radius 1, FilterManager.cpp:16) collapses to the centre tap:
    # mitchell(2*1/1) = 0 for the ring, so only (0,0) passes the `weight > eps` test
    t = np.zeros(9, dtype=np.float32)
    lib.orc_filter_table(abi.FILTER_MITCHELL, 1, t.ctypes.data_as(C.POINTER(C.c_float)))
    assert (t > np.finfo(np.float32).eps).sum() == 1 and abs(t[4] - 1) < 1e-6


def test_mjitt_permute_is_bijection():
    lib = ob.load()
    for l in (1, 2, 7, 16, 100, 1024):
        for p in (1, 0xdeadbeef, 14512080):
            assert sorted(lib.orc_mjitt_permute(i, l, p) for i in range(l)) == list(range(l))
