"""Known answers the reference's own unit tests hold for two helpers of the hot path's periphery, applied to the checker's restatements:
Spherical::uv_from_normal / cartesian_from_uv (src/tests/sphere.cpp:12-67; the textured environment light looks its image up through
them, the sphere light samples through them) and BoundingBox::intersectsRange / intersects (src/tests/boundingbox.cpp:128-280; the
quadric callbacks clip their roots with it)."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob

PRT_EPSILON = 2 * 1.1920929e-07   # Test.h:224-225
F3 = C.c_float * 3


def _uv_round_trip(u, v):
    lib = ob.load()
    n, uv = F3(), (C.c_float * 2)()
    lib.orc_cartesian_from_uv(u, v, n)
    lib.orc_uv_from_normal(n, uv)
    return tuple(n), (uv[0], uv[1])


@pytest.mark.parametrize("uv,expect", [((0.0, 0.0), (0.0, 0.0)), ((0.5, 0.5), (0.5, 0.5)), ((0.75, 0.25), (0.75, 0.25)), ((0.25, 0.75), (0.25, 0.75))])
def test_spherical_uv_round_trips_of_the_reference(uv, expect):
    """sphere.cpp "UV (0,0)", "(0.5,0.5)", "(0.75,0.25)", "(0.25,0.75)": uv_from_normal(cartesian_from_uv(u, v)) == (u, v) to PRT_EPSILON --
    met by the shared fp32 atan2 / acos / sincos forms (not libm) to one ulp of the result more (3e-7 instead of 2.4e-7 at worst)."""
    n, got = _uv_round_trip(*uv)
    assert abs(np.linalg.norm(n) - 1) < 1e-6
    assert abs(got[0] - expect[0]) <= PRT_EPSILON + 6e-8 and abs(got[1] - expect[1]) <= PRT_EPSILON + 6e-8, got


@pytest.mark.parametrize("uv,expect_v", [((1.0, 0.0), 0.0), ((0.0, 1.0), 1.0), ((1.0, 1.0), 1.0)])
def test_spherical_uv_at_the_poles(uv, expect_v):
    """sphere.cpp "UV (1,0)", "(0,1)", "(1,1)" ('Ambiguous' in the reference's own words: at a pole u is whatever the sign of libm's sin(pi)
    makes of atan2): v is exact; u is 0 or 0.5 there, 0 here at v = 0 as the reference expects, and at v = 1 either value names the same
    direction -- the checker's quadrant-reduced sine returns an exact 0 at pi, so it takes the x = 1e-5 branch of Spherical.h:12 (u = 0)
    where libm's -8.7e-8 leads to u = 0.5."""
    n, got = _uv_round_trip(*uv)
    assert abs(got[1] - expect_v) <= PRT_EPSILON
    assert abs(n[2] - (1 - 2 * expect_v)) <= PRT_EPSILON and abs(n[0]) < 1e-6 and abs(n[1]) < 1e-6
    assert min(abs(got[0] - 0.0), abs(got[0] - 0.5)) <= PRT_EPSILON
    if expect_v == 0.0:
        assert abs(got[0]) <= PRT_EPSILON


BOX = ((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))    # BoundingBox(2, 2, 2): width, height, depth around the origin


def _range(o, d):
    lib = ob.load()
    r = (C.c_float * 2)()
    ok = lib.orc_box_range(F3(*BOX[0]), F3(*BOX[1]), F3(*o), F3(*d), r)
    return ok, r[0], r[1]


def test_box_range_known_answer_of_the_reference():
    """boundingbox.cpp "Intersects Range": Entry 1, Exit 3, Successful."""
    assert _range((-2, 0, 0), (1, 0, 0)) == (1, 1.0, 3.0)


@pytest.mark.parametrize("o,d", [((-2, 0, 0), (1, 0, 0)), ((2, 0, 0), (-1, 0, 0)), ((0, 0, -2), (0, 0, 1)), ((0, 0, 2), (0, 0, -1)),
                                 ((0, -2, 0), (0, 1, 0)), ((0, 2, 0), (0, -1, 0)), ((1, 2, 0), (-1, -1, 0))])
def test_box_entry_distances_of_the_reference(o, d):
    """boundingbox.cpp "Intersects Left / Right / Front / Back / Bottom / Top / Complex": a ray from outside meets the box at distance 1
    (BoundingBox::intersects reports the range's entry, BoundingBox.cpp:25-47) -- axis-parallel rays included (their 1 / 0 slabs)."""
    ok, entry, exit_ = _range(o, d)
    assert ok == 1 and entry == 1.0 and exit_ > entry
    hit = np.array(o, np.float32) + np.float32(entry) * np.array(d, np.float32)
    assert np.abs(hit).max() == 1.0                       # on the surface


@pytest.mark.parametrize("d", [(1, 0, 0), (0, 0, 1), (0, 1, 0)])
def test_box_exit_distance_from_inside(d):
    """boundingbox.cpp "Intersects Right / Back / Top Inside": from the centre the reported distance is the exit, 1; the entry is clamped
    to the ray's MinT = PR_EPSILON (Ray.h:25)."""
    ok, entry, exit_ = _range((0, 0, 0), d)
    assert ok == 1 and exit_ == 1.0 and entry == np.float32(1.1920929e-07)
