"""Textured environment light (environment.cpp with an image radiance: Distribution2D over the map, 1 / (2 pi^2 sin theta) Jacobian):
closed-form and consistency checks of the CPU checker, and the HIP path against it."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from pearray_amd import _cabi as abi
from pearray_amd import scene
from test_oracle_skysun import WL, light_eval, light_sample


def two_blob_image(h=16, w=32):
    """A dark sky with two bright patches: one near the nadir (rows 2-3: row 0 is the nadir, the last row the zenith -- the lookup is
    row = (1 - v) * h with v = theta / pi, OIIO's t = 1 - v of environment.cpp:53-101), a dimmer reddish one in rows 9-10 on the other side."""
    rgb = np.full((h, w, 3), 0.02, np.float32)
    rgb[2:4, 5:9] = (0.9, 0.9, 0.8)
    rgb[9:11, 20:24] = (0.8, 0.3, 0.1)
    return rgb


def env_scene(size=32, spp=64, distribution=True, background=None, compensation=False, image=None, transform=scene.IDENTITY, glass=False, **settings):
    b = scene.SceneBuilder(size, size)
    b.settings.aa_sampler, b.settings.aa_samples = abi.SAMPLER_RANDOM, spp
    for k, v in settings.items():
        setattr(b.settings, k, v)
    T = np.array([[1, 0, 0, 0], [0, 0.6, 0.8, 2.4], [0, -0.8, 0.6, 1.8], [0, 0, 0, 1]], dtype=np.float32)
    b.set_camera(T, width=0.8, height=0.8, near=0.01, far=100, local_direction=(0, 0, -1), local_up=(0, 1, 0), local_right=(1, 0, 0))
    floor = b.lambert(b.refl(0.7, 0.7, 0.7))
    b.add_mesh([[-4, -4, 0], [4, -4, 0], [4, 4, 0], [-4, 4, 0]], [[0, 1, 2, 3]], floor, normals=[[0, 0, 1]] * 4)
    box = b.dielectric(b.lookup_index("bk7")) if glass else b.lambert(b.refl(0.2, 0.5, 0.7))
    P = [[-0.4, -0.4, 0], [0.4, -0.4, 0], [0.4, 0.4, 0], [-0.4, 0.4, 0], [-0.4, -0.4, 0.8], [0.4, -0.4, 0.8], [0.4, 0.4, 0.8], [-0.4, 0.4, 0.8]]
    F = [[0, 3, 2, 1], [4, 5, 6, 7], [0, 1, 5, 4], [1, 2, 6, 5], [2, 3, 7, 6], [3, 0, 4, 7]]
    b.add_mesh(P, F, box)
    img = b.rgb_image_to_coefficients(two_blob_image() if image is None else image)
    b.environment_light(b.smul(b.illuminant_d65(), b.spectrum_const(2.0)), background=None if background is None else b.spectrum_const(background),
                        image=img, distribution=distribution, compensation=compensation, transform=transform)
    return b.build()


def test_samples_carry_the_pdf_and_radiance_that_eval_reports_for_their_direction():
    o = ob.OracleScene(env_scene())
    rng = np.random.default_rng(3)
    agree = 0
    n = 400
    for _ in range(n):
        u0, u1 = float(rng.random()), float(rng.random())
        L, pdf, rad = light_sample(o, 0, u0, u1)
        assert abs(np.linalg.norm(L) - 1) < 1e-5 and pdf > 0 and np.isfinite(rad).all()
        erad, epdf = light_eval(o, 0, L)
        agree += int(abs(epdf - pdf) <= 2e-3 * pdf and np.allclose(erad, rad, rtol=1e-4, atol=1e-7))
    assert agree >= 0.97 * n     # the rest sit on a texel edge, where direction -> uv lands in the neighbour


def test_importance_sampled_integral_equals_the_sum_over_texels():
    """E[radiance / pdf] over the light's own samples = integral of the radiance over the sphere = sum over texels x their solid angle."""
    sc = env_scene()
    o = ob.OracleScene(sc)
    rng = np.random.default_rng(5)
    est = np.mean([r[0] / p for _, p, r in (light_sample(o, 0, float(rng.random()), float(rng.random())) for _ in range(6000))])
    h, w = 16, 32
    total = 0.0
    for row in range(h):       # row 0 is the NADIR: v = 1 - (row + 0.5) / h, theta = pi * v ... the lookup flips v (t = 1 - v)
        v = 1 - (row + 0.5) / h
        for col in range(w):
            th, ph = np.pi * v, 2 * np.pi * (col + 0.5) / w
            d = (np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th))
            rad, _ = light_eval(o, 0, d)
            total += rad[0] * (np.cos(np.pi * (v - 0.5 / h)) - np.cos(np.pi * (v + 0.5 / h))) * (2 * np.pi / w)
    assert abs(est - total) < 0.03 * total


def test_next_event_estimation_and_bsdf_sampling_agree_on_the_lit_floor():
    def mean_y(**kw):
        o = ob.OracleScene(env_scene(size=20, spp=300, mapper=abi.MAPPER_RANDOM, **kw))
        o.render(300, threads=8)
        xyz, _, fb = o.output()
        assert np.isfinite(xyz).all() and not fb.any()
        return float(xyz[..., 1].mean())
    both, bsdf_only, hemi = mean_y(), mean_y(nee=0), mean_y(distribution=False)
    assert abs(both - bsdf_only) < 0.04 * bsdf_only
    # without the distribution the reference looks the radiance of a light sample up at the RANDOM NUMBERS, not at the sampled direction
    # (environment.cpp:88-99): a different (wrong) estimate -- kept, and only required to be finite here
    assert np.isfinite(hemi) and hemi > 0


def test_power_is_the_mean_over_the_reference_uv_grid_and_bad_images_are_refused():
    sc = env_scene()
    o = ob.OracleScene(sc)
    pw = (C.c_float * 4)()
    o.lib.orc_inf_light_power(o.h, 0, WL, pw)
    vals = []
    for i in range(1024):      # NodeUtils::average: texture coordinates (x / 32, y / 32) of the Morton index i
        x = sum(((i >> (2 * k)) & 1) << k for k in range(5)); y = sum(((i >> (2 * k + 1)) & 1) << k for k in range(5))
        u, v = x / 32.0, y / 32.0
        th, ph = np.pi * min(max(v, 1e-4), 1 - 1e-4), 2 * np.pi * u + 1e-4
        vals.append(light_eval(o, 0, (np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)))[0][0])
    assert abs(pw[0] - np.mean(vals)) < 0.05 * np.mean(vals)
    bad = env_scene()
    bad.desc.lights[0].table_offset = bad.desc.n_spectral_table_values - 5
    with pytest.raises(RuntimeError):
        ob.OracleScene(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(distribution=False), dict(background=0.3, glass=True), dict(compensation=True),
                                dict(transform=np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32), mis=abi.MIS_POWER)])
def test_gpu_textured_environment_is_bit_exact(kw):
    import test_gpu_parity as tg
    g, o = tg.render_both(env_scene(size=64, spp=6, **kw), iters=6)
    tg.assert_parity(g, o, exact=True)
    st = g.statistics()
    assert st["shadow_rays"] > 0 and st["background_hits"] > 0


@pytest.mark.gpu
def test_gpu_textured_environment_in_every_pipeline(monkeypatch):
    import test_gpu_parity as tg
    sc = env_scene(size=48, spp=4, glass=True)
    ref = tg._render_mode(monkeypatch, "lockstep", sc, [4])
    for mode in ("streaming", "persistent"):
        out = tg._render_mode(monkeypatch, mode, sc, [4])
        for a, b in zip(ref[0] + ref[1], out[0] + out[1]):
            assert np.array_equal(a, b), mode


def test_orientation_of_the_map_last_row_is_the_zenith():
    """Orientation, which the integrals above cannot see: radiance towards +z comes from the LAST row of the image, towards -z from row 0
    (row = (1 - v) * h, v = theta / pi), and a bright last row lights an upward-facing floor where a bright first row does not."""
    def sky(bright_row):
        rgb = np.full((8, 16, 3), 0.01, np.float32)
        rgb[bright_row] = (0.9, 0.9, 0.9)
        return rgb
    up, down = ob.OracleScene(env_scene(image=sky(7))), ob.OracleScene(env_scene(image=sky(0)))
    assert light_eval(up, 0, (0.05, 0.0, 0.99875))[0][0] > 20 * light_eval(up, 0, (0.05, 0.0, -0.99875))[0][0]
    assert light_eval(down, 0, (0.05, 0.0, -0.99875))[0][0] > 20 * light_eval(down, 0, (0.05, 0.0, 0.99875))[0][0]

    def floor_y(image):
        o = ob.OracleScene(env_scene(size=16, spp=64, mapper=abi.MAPPER_RANDOM, image=image))
        o.render(64, threads=8)
        return float(o.output()[0][..., 1].mean())
    assert floor_y(sky(7)) > 5 * floor_y(sky(0))
