"""Infinite lights in the CPU checker (environment.cpp untextured, distant.cpp, direct.cpp:415-456 handleInfLights,
LightSampler.cpp:62-71): closed-form checks.  CPU only."""
import ctypes as C

import numpy as np

import oracle_binding as ob
from pearray_amd import _cabi as abi
from pearray_amd import scene


def floor_scene(albedo=0.5, spp=128, look_down=True, lights=("env",), size=24, **settings):
    b = scene.SceneBuilder(size, size)
    b.settings.aa_sampler, b.settings.aa_samples = abi.SAMPLER_RANDOM, spp
    b.settings.mapper = abi.MAPPER_RANDOM            # uniform wavelengths: Y of a flat spectrum is the same on every path
    for k, v in settings.items():
        setattr(b.settings, k, v)
    T = np.eye(4, dtype=np.float32); T[:3, 3] = [0, 0, 2]
    b.set_camera(T, width=0.2, height=0.2, near=0.01, far=100, local_direction=(0, 0, -1 if look_down else 1), local_up=(0, 1, 0), local_right=(1, 0, 0))
    m = b.lambert(b.spectrum_const(albedo))
    b.add_mesh([[-50, -50, 0], [50, -50, 0], [50, 50, 0], [-50, 50, 0]], [[0, 1, 2, 3]], m, normals=[[0, 0, 1]] * 4)
    for l in lights:
        if l == "env":
            b.environment_light(b.spectrum_const(1.0))
        elif l == "env_split":
            b.environment_light(b.spectrum_const(1.0), background=b.spectrum_const(3.0))
        elif l == "sun":
            b.distant_light(b.spectrum_const(np.pi), direction=(0.0, 0.6, 0.8))
    return b.build()


def mean_y(sc, iters):
    o = ob.OracleScene(sc)
    o.render(iters, threads=8)
    xyz, smp, fb = o.output()
    assert np.isfinite(xyz).all() and (fb == 0).all()
    return float(xyz[..., 1].mean()), o.statistics()


def unit_y():
    """Y of unit radiance on a regular path.  Camera rays that leave the scene are splatted by handleBackgroundGroup with
    weight Ones() for all four wavelengths (IntegratorUtils.h:38) instead of 1/4 each like every other fragment
    (direct.cpp:384,430: heroFactor / heroFactor.sum()), so the directly visible background is four times brighter."""
    sky, _ = mean_y(floor_scene(look_down=False, spp=16), 16)
    return sky / 4


def test_white_furnace_under_a_constant_environment():
    """A Lambert plane of albedo a under unit radiance from the whole upper hemisphere reflects a (NEE + BSDF hits of the
    background, MIS-combined): the estimate must be unbiased."""
    sky = unit_y()
    for a in (0.5, 0.9):
        y, st = mean_y(floor_scene(albedo=a), 128)
        assert abs(y / sky - a) < 0.02 * a, (y, sky, a)
        assert st["background_hits"] > st["pixel_samples"]            # NEE samples of an infinite light count as background hits
    y_nonee, _ = mean_y(floor_scene(albedo=0.5, nee=0), 128)         # BSDF sampling alone must agree
    assert abs(y_nonee / sky - 0.5) < 0.02


def test_split_environment_shows_the_background_only_to_camera_rays():
    sky, _ = mean_y(floor_scene(look_down=False, spp=16, lights=("env_split",)), 16)
    plain, _ = mean_y(floor_scene(look_down=False, spp=16), 16)
    assert abs(sky / plain - 3.0) < 1e-3                              # :background 3 vs radiance 1
    lit, _ = mean_y(floor_scene(albedo=0.5, lights=("env_split",)), 128)
    assert abs(lit / unit_y() - 0.5) < 0.02                           # the floor is lit by :radiance


def test_distant_light_is_a_delta_reached_only_by_nee():
    """Irradiance E = pi from direction d onto a Lambert plane: L = a / pi * E * cos(theta) = a * 0.8."""
    sky = unit_y()
    y, st = mean_y(floor_scene(albedo=0.5, lights=("sun",)), 64)
    assert abs(y / sky - 0.5 * 0.8) < 0.01
    dark, _ = mean_y(floor_scene(albedo=0.5, lights=("sun",), nee=0), 8)
    assert dark == 0.0
    # Reference quirk kept on purpose: a delta light's NEE pdf is 1 WITHOUT its selection probability (direct.cpp:291-304 applies
    # `lightPdfS *= lsample.second` only in the non-delta branch), so next to another light it is dimmed by p_select.
    # Intensities are 2 pi R mean(power): environment 1, sun pi  ->  p_sun = pi / (1 + pi).
    both, _ = mean_y(floor_scene(albedo=0.5, lights=("env", "sun")), 256)
    assert abs(both / sky - (0.5 + 0.4 * np.pi / (1 + np.pi))) < 0.02


def test_light_selection_distribution_appends_infinite_lights():
    """LightSampler.cpp:62-71: intensity of an infinite light = 2 pi R mean(power), R = origin-centred bounding-sphere radius."""
    def selector(sc):
        o = ob.OracleScene(sc)
        n, cdf, inten = C.c_uint32(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        o.lib.orc_light_selector(o.h, C.byref(n), C.byref(cdf), C.byref(inten))
        return n.value, np.ctypeslib.as_array(cdf, shape=(n.value + 1,)).copy(), np.ctypeslib.as_array(inten, shape=(n.value,)).copy()
    n, cdf, _ = selector(scene.cornell_box(8, 8, spp=1))
    assert n == 1 and cdf.tolist() == [0.0, 1.0]
    b = scene.SceneBuilder(8, 8)
    b.set_camera(scene.IDENTITY)
    m = b.lambert(b.spectrum_const(0.5))
    b.add_mesh([[-3, -4, 0], [3, -4, 0], [3, 4, 0], [-3, 4, 1]], [[0, 1, 2, 3]], m)     # |hi| = |(3,4,1)|, |lo| = |(-3,-4,0)| = 5
    b.environment_light(b.spectrum_const(1.0))
    b.distant_light(b.spectrum_const(3.0))
    n, cdf, inten = selector(b.build())
    assert n == 2
    assert np.allclose(cdf, [0, 0.25, 1.0], atol=1e-6) and np.allclose(inten, [0.25, 0.75], atol=1e-6)   # powers 1 : 3, same 2 pi R factor
