"""Spherical and fisheye cameras (plugins/main/cameras/spherical.cpp, fisheye.cpp) and the CIE sky lights (infinitelights/cie_sky.cpp) in
the CPU checker: closed forms in float64 next to the fp32 restatement, the clipped fisheye sample (no camera ray, RenderTile.cpp:71-131 /
StreamPipeline.cpp:104-105), and the `.prc` loader's view of the two reference examples that use them.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_binding as ob
from pearray_amd import _cabi as abi
from pearray_amd import scene

from test_oracle_skysun import WL, floor_under, light_eval, light_sample
from test_oracle_inflights import mean_y

REF_EXAMPLES = "/root/reference/examples"


def camera_scene(setup, width=32, height=16, spp=4):
    b = scene.SceneBuilder(width, height)
    b.settings.aa_sampler, b.settings.aa_samples = abi.SAMPLER_RANDOM, spp
    b.settings.mapper = abi.MAPPER_RANDOM
    T = np.eye(4, dtype=np.float32); T[:3, 3] = [0.5, -1.0, 2.0]
    setup(b, T)
    m = b.lambert(b.spectrum_const(0.5))
    b.add_mesh([[-50, -50, 0], [50, -50, 0], [50, 50, 0], [-50, 50, 0]], [[0, 1, 2, 3]], m, normals=[[0, 0, 1]] * 4)
    b.environment_light(b.spectrum_const(1.0))
    return b.build()


def ray(o, px, py):
    org, d = (C.c_float * 3)(), (C.c_float * 3)()
    ok = o.lib.orc_camera_ray(o.h, px, py, 0.0, 0.0, org, d)
    return ok, np.array(org[:]), np.array(d[:], dtype=np.float64)


def test_spherical_camera_follows_the_closed_form():
    frame = dict(local_direction=(0, 1, 0), local_right=(1, 0, 0), local_up=(0, 0, 1))
    W, H = 32, 16
    sc = camera_scene(lambda b, T: b.set_spherical_camera(T, theta_start=-1.570796, near=0.01, far=100, **frame), W, H)
    o = ob.OracleScene(sc)
    rng = np.random.default_rng(5)
    for _ in range(300):
        px, py = float(rng.uniform(-0.5, W - 0.5)), float(rng.uniform(-0.5, H - 0.5))
        ok, org, d = ray(o, px, py)
        assert ok == 1 and np.allclose(org, [0.5, -1.0, 2.0])
        nx, ny = px / W, 1 - py / H
        theta = -1.570796 + ny * (np.pi / 2 + 1.570796)
        phi = -np.pi + nx * (2 * np.pi)
        # fromTangentSpace(Up, Right, Direction, (sin phi cos theta, cos phi cos theta, sin theta)): N = up, Nx = right, Ny = direction
        want = np.array([0, 0, 1.0]) * np.sin(theta) + np.array([0, 1.0, 0]) * (np.cos(phi) * np.cos(theta)) + np.array([1.0, 0, 0]) * (np.sin(phi) * np.cos(theta))
        assert abs(np.linalg.norm(d) - 1) < 1e-6 and np.allclose(d, want, atol=3e-6)
    # the image centre looks along the local direction at the middle elevation (here the horizon)
    _, _, d = ray(o, W / 2, H / 2)
    assert np.allclose(d, [0, 1, 0], atol=1e-5)


@pytest.mark.parametrize("map_type", [abi.FISHEYE_CIRCULAR, abi.FISHEYE_CROPPED, abi.FISHEYE_FULL])
@pytest.mark.parametrize("size", [(32, 16), (16, 32), (24, 24)])
def test_fisheye_camera_follows_the_closed_form_and_clips_outside_the_unit_circle(map_type, size):
    W, H = size
    fov = float(np.float32(np.deg2rad(160.0)))
    frame = dict(local_direction=(0, 0, 1), local_right=(1, 0, 0), local_up=(0, 1, 0))
    sc = camera_scene(lambda b, T: b.set_fisheye_camera(T, fov=fov, map_type=map_type, clip_range=True, near=0.01, far=100, **frame), W, H)
    sc_open = camera_scene(lambda b, T: b.set_fisheye_camera(T, fov=fov, map_type=map_type, clip_range=False, near=0.01, far=100, **frame), W, H)
    o, o_open = ob.OracleScene(sc), ob.OracleScene(sc_open)
    aspect = W / H
    if map_type == abi.FISHEYE_CIRCULAR:
        xa, ya = (1 if aspect < 1 else aspect), (1 if aspect > 1 else aspect)
    elif map_type == abi.FISHEYE_CROPPED:
        xa, ya = (1 / aspect if aspect < 1 else 1), (1 / aspect if aspect > 1 else 1)
    else:
        f = np.sqrt(aspect * aspect + 1) * H / min(W, H)
        xa, ya = (1 if aspect < 1 else 1 / aspect) * f, (1 if aspect > 1 else aspect) * f
    rng = np.random.default_rng(7)
    clipped = 0
    for _ in range(400):
        px, py = float(rng.uniform(-0.5, W - 0.5)), float(rng.uniform(-0.5, H - 0.5))
        nx, ny = 2 * (px / W - 0.5) / xa, -(2 * (py / H - 0.5) / ya)
        r = np.hypot(nx, ny)
        ok, _, d = ray(o, px, py)
        ok_open, _, d_open = ray(o_open, px, py)
        assert ok_open == 1
        if abs(r - 1) > 1e-5:
            assert ok == (1 if r <= 1 else 0)
        clipped += 1 - ok
        theta = r * fov / 2
        want = np.array([nx / r * np.sin(theta), ny / r * np.sin(theta), np.cos(theta)]) if r > 1e-6 else np.array([0, 0, 1.0])
        assert np.allclose(d_open, want, atol=3e-6)
        if ok:
            assert np.array_equal(d, d_open)
    if map_type == abi.FISHEYE_FULL:
        assert clipped == 0      # the image circle circumscribes the sensor
    else:
        assert clipped > 0
    _, _, d = ray(o, W / 2, H / 2)
    assert np.allclose(d, [0, 0, 1], atol=1e-6)


def test_clipped_fisheye_samples_are_counted_but_trace_and_splat_nothing():
    W = H = 24
    frame = dict(local_direction=(0, 0, -1), local_right=(1, 0, 0), local_up=(0, 1, 0))
    make = lambda clip: camera_scene(lambda b, T: b.set_fisheye_camera(T, clip_range=clip, near=0.01, far=100, **frame), W, H, spp=8)  # noqa: E731
    o, o_open = ob.OracleScene(make(True)), ob.OracleScene(make(False))
    o.render(8); o_open.render(8)
    st, st_open = o.statistics(), o_open.statistics()
    assert st["pixel_samples"] == st_open["pixel_samples"] == W * H * 8
    assert st["primary_rays"] < st_open["primary_rays"] == W * H * 8
    xyz, samples, _ = o.output()
    xyz_open, samples_open, _ = o_open.output()
    yy, xx = np.mgrid[0:H, 0:W]
    r_max = np.hypot((np.abs(xx + 0.5 - W / 2) + 0.5) / (W / 2), (np.abs(yy + 0.5 - H / 2) + 0.5) / (H / 2))   # farthest point of the pixel
    r_min = np.hypot(np.maximum(np.abs(xx + 0.5 - W / 2) - 0.5, 0) / (W / 2), np.maximum(np.abs(yy + 0.5 - H / 2) - 0.5, 0) / (H / 2))
    inside, outside = r_max < 0.97, r_min > 1.03
    assert outside.sum() > 20 and inside.sum() > 200
    assert np.all(xyz[outside] == 0) and np.all(samples[outside] == 0)
    assert np.all(xyz_open[outside].sum(axis=-1) > 0)
    # a pixel entirely inside the circle sees the same samples either way: identical values
    assert np.array_equal(xyz[inside], xyz_open[inside]) and np.array_equal(samples[inside], samples_open[inside])


def cie_scene(cloudy, ground_tint=None, brightness=0.2, transform=scene.IDENTITY, **kw):
    def light(b):
        zen = b.spectrum_table(390.0, 830.0, np.linspace(0.6, 1.4, 12))
        gt = None if ground_tint is None else b.spectrum_const(ground_tint)
        b.cie_sky_light(zen, ground_tint=gt, ground_brightness=brightness, cloudy=cloudy, transform=transform)
    return floor_under(light, **kw)


@pytest.mark.parametrize("cloudy", [False, True])
@pytest.mark.parametrize("ground_tint", [None, 0.3])
def test_cie_sky_radiance_pdf_and_sampling_follow_the_closed_form(cloudy, ground_tint):
    o = ob.OracleScene(cie_scene(cloudy, ground_tint, brightness=0.35))
    zen = np.interp(np.array(WL[:], dtype=np.float64), np.linspace(390.0, 830.0, 12), np.linspace(0.6, 1.4, 12))
    gnd = zen if ground_tint is None else np.full(4, ground_tint)
    rng = np.random.default_rng(11)

    def want(z):
        a = (z + 1.01) ** 10
        b = 1 / a
        c1, c2 = ((1 + 2 * z) / 3, 0.7777777) if cloudy else (1.0, 1.0)
        return (zen * (c1 * a) + gnd * (0.35 * c2 * b)) / (a + b)
    for _ in range(200):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        rad, pdf = light_eval(o, 0, d)
        assert np.allclose(rad, want(d[2]), rtol=3e-5, atol=1e-6)
        assert abs(pdf - abs(d[2]) / np.pi) < 1e-6
        L, pdf_s, rad_s = light_sample(o, 0, float(rng.uniform()), float(rng.uniform()))
        assert L[2] >= 0 and abs(np.linalg.norm(L) - 1) < 1e-5 and abs(pdf_s - L[2] / np.pi) < 1e-6
        assert np.allclose(rad_s, want(float(L[2])), rtol=3e-5, atol=1e-6)
    # straight up: the zenith tint (the ground term is (1 / 2.01^10)^2 of it); straight down: the ground term
    rad, _ = light_eval(o, 0, (0, 0, 1))
    assert np.allclose(rad, zen if not cloudy else zen * 1.0, rtol=1e-3)
    rad, _ = light_eval(o, 0, (0, 0, -1))
    assert np.allclose(rad, gnd * 0.35 * (0.7777777 if cloudy else 1.0), rtol=1e-3) or cloudy   # cloudy: c1 < 0 below the horizon shifts it slightly
    # power() is the radiance towards +z of the WORLD (cie_sky.cpp:80)
    pw = (C.c_float * 4)()
    o.lib.orc_inf_light_power(o.h, 0, WL, pw)
    assert np.allclose(pw[:], want(1.0), rtol=3e-5)


def test_cie_sky_follows_the_light_transform():
    R = np.eye(4, dtype=np.float32)
    R[:3, :3] = [[1, 0, 0], [0, 0, 1], [0, -1, 0]]     # local +z -> world +y
    o = ob.OracleScene(cie_scene(False, 0.0, transform=R))
    up, _ = light_eval(o, 0, (0, 1, 0))
    side, _ = light_eval(o, 0, (0, 0, 1))
    down, _ = light_eval(o, 0, (0, -1, 0))
    assert np.allclose(side / up, 1.01 ** 10 / (1.01 ** 10 + 1.01 ** -10), rtol=1e-4) and np.all(down < 1e-6 * up)


def test_uniform_cie_sky_over_a_floor_is_close_to_a_constant_environment():
    """A ground-less uniform sky is ~constant over the upper hemisphere ((z + 1.01)^20 / (1 + (z + 1.01)^20) >= 0.55 at the horizon, ~1 above
    z = 0.2): a Lambert floor of albedo 0.5 under radiance 1 reflects a little less than 0.5."""
    def light(b):
        b.cie_sky_light(b.spectrum_const(1.0), ground_tint=b.spectrum_const(0.0))
    y, _ = mean_y(floor_under(light, spp=128), 128)
    env, _ = mean_y(floor_under(lambda b: b.environment_light(b.spectrum_const(1.0)), spp=128), 128)
    assert 0.85 * env < y < env


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="reference examples not present")
def test_reference_examples_with_spherical_and_fisheye_cameras_load():
    p = scene.PrcScene(path=os.path.join(REF_EXAMPLES, "sky.prc"))
    cam = p.desc.camera
    assert cam.kind == abi.CAMERA_SPHERICAL and abs(cam.theta_start + 1.570796) < 1e-6 and abs(cam.theta_end - np.pi / 2) < 1e-6
    kinds = sorted(p.desc.lights[i].kind for i in range(p.desc.n_lights))
    assert kinds == [abi.LIGHT_SUN, abi.LIGHT_CIE_SKY] and any(p.desc.lights[i].flags & abi.SKYF_CLOUDY for i in range(p.desc.n_lights))
    q = scene.PrcScene(path=os.path.join(REF_EXAMPLES, "skylens.prc"))
    cam = q.desc.camera
    assert cam.kind == abi.CAMERA_FISHEYE and cam.clip_range == 1 and cam.fisheye_map == abi.FISHEYE_CIRCULAR
    assert abs(cam.fov - np.float32(180.0) * (np.float32(np.pi) / np.float32(180.0))) < 1e-6
    assert q.desc.n_lights == 1 and q.desc.lights[0].kind == abi.LIGHT_SKY
