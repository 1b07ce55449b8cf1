"""Randomised parity: scenes nobody designed.  A seeded generator throws together geometry (boxes of random pose and size, loose triangles,
plane / sphere / quadric entities), materials of every closure the path knows (Lambert, mirror, smooth and rough dielectric and conductor,
principled), lights (an emissive panel, now and then a second emitter, plus any of plain or textured environment, distant, sun, CIE sky, Hosek sky), one of the four cameras, and render settings
(sampler, spectral mapper, MIS, NEE, hero wavelengths, depth limits, pixel filter) -- the HIP path must reproduce the checker on every one:
primary hit ids, sample and feedback planes, the eleven statistics, and the frame bit for bit (single-tap filters) or to 1e-5 (multi-tap)."""
import os

import numpy as np
import pytest

import oracle_binding as ob
from pearray_amd import _cabi as abi
from pearray_amd import backend, scene

pytestmark = pytest.mark.gpu


def _rot(rng):
    a, b, c = rng.uniform(0, 2 * np.pi, 3)
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    return rx @ ry @ rz


def _xf(rng, pos, scale):
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = (_rot(rng) * np.asarray(scale, dtype=np.float64)).astype(np.float32)
    m[:3, 3] = pos
    return m


BOX_P = [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]]
BOX_F = [[0, 3, 2, 1], [4, 5, 6, 7], [0, 1, 5, 4], [2, 3, 7, 6], [1, 2, 6, 5], [0, 4, 7, 3]]


def _material(b, rng, textured=False):
    kind = rng.integers(0, 9)
    plain = lambda lo=0.2, hi=0.9: b.refl(*rng.uniform(lo, hi, 3))   # noqa: E731
    # with texture coordinates around, reflectances may be checkerboards of two colours (CheckerboardNode: unscaled, isotropic, anisotropic)
    c = (lambda lo=0.2, hi=0.9: b.checkerboard(plain(lo, hi), plain(lo, hi), *([None, None], [4.0, None], [3.0, 7.0])[int(rng.integers(3))])) if textured else plain   # noqa: E731
    if kind == 0 or kind == 1:
        return b.lambert(c())
    if kind == 2:
        return b.mirror(specularity=c(0.6, 0.95))
    if kind == 3:
        return b.dielectric(b.lookup_index("bk7") if rng.integers(2) else b.spectrum_const(float(rng.uniform(1.2, 1.8))), thin=bool(rng.integers(4) == 0))
    if kind == 4:
        return b.conductor() if rng.integers(2) else b.conductor(eta=b.spectrum_const(float(rng.uniform(0.2, 1.5))), k=b.spectrum_const(float(rng.uniform(2, 4))))
    if kind == 5:
        return b.rough_conductor(float(rng.uniform(0.05, 0.6)), roughness_y=float(rng.uniform(0.05, 0.6)) if rng.integers(2) else None, vndf=bool(rng.integers(2)))
    if kind == 6:
        return b.rough_dielectric(float(rng.uniform(0.05, 0.5)), ior=b.spectrum_const(float(rng.uniform(1.3, 1.7))), vndf=bool(rng.integers(2)))
    return b.principled(base=c(), roughness=float(rng.uniform(0.1, 0.9)), vndf=bool(rng.integers(2)))


def random_scene(seed, width=56, height=40):
    rng = np.random.default_rng(seed)
    b = scene.SceneBuilder(width, height)
    s = b.settings
    s.aa_sampler = int(rng.choice([abi.SAMPLER_RANDOM, abi.SAMPLER_MJITT, abi.SAMPLER_SOBOL, abi.SAMPLER_HALTON, abi.SAMPLER_HAMMERSLEY, abi.SAMPLER_UNIFORM, abi.SAMPLER_STRATIFIED]))
    s.aa_samples = max(4, int(os.environ.get("PRGPU_TEST_RANDOM_ITERS", "4")))
    s.mapper = int(rng.choice([abi.MAPPER_SPD_CMIS, abi.MAPPER_SPD_CMIS, abi.MAPPER_RANDOM, abi.MAPPER_SPD_HERO, abi.MAPPER_CIE, abi.MAPPER_AGH_CMIS]))
    s.mis = int(rng.integers(2))
    s.nee = int(rng.integers(5) != 0)
    s.spectral_hero = int(rng.integers(4) != 0)
    s.max_ray_depth = int(rng.choice([3, 8, 64]))
    s.soft_max_ray_depth = min(s.max_ray_depth, int(rng.choice([1, 4])))
    s.seed = int(rng.integers(1, 1 << 30))
    multi_tap = rng.integers(4) == 0
    s.filter, s.filter_radius = (int(rng.choice([abi.FILTER_GAUSSIAN, abi.FILTER_TRIANGLE, abi.FILTER_LANCZOS])), int(rng.integers(1, 3))) if multi_tap else (abi.FILTER_BLOCK, 0)
    # a floor, an emissive panel, then whatever the dice say
    floor = b.lambert(b.refl(0.6, 0.6, 0.6))
    light = rng.integers(0, 7)   # 0: the panel alone; 1 .. 6: plus an infinite light
    if light == 0 or rng.integers(2):   # a closed room (always when the panel is the only light): paths bounce until roulette ends them
        room = np.eye(4, dtype=np.float32); room[0, 0] = room[1, 1] = 8.5; room[2, 2] = 3.0; room[2, 3] = 3.0
        b.add_mesh(BOX_P, BOX_F if light == 0 else BOX_F[:1] + BOX_F[2:], b.lambert(b.refl(*rng.uniform(0.4, 0.8, 3))), transform=room)   # (open to the sky: no ceiling)
    else:
        b.add_mesh([[-6, -6, 0], [6, -6, 0], [6, 6, 0], [-6, 6, 0]], [[0, 1, 2, 3]], floor)
    lamp = b.diffuse_emission(b.smul(b.illuminant_d65(), b.illum(*rng.uniform(3, 12, 3))))
    b.add_mesh([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], [[0, 3, 2, 1]], floor, emission=lamp, transform=_xf(rng, [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(3, 4.5)], rng.uniform(0.4, 1.2)))
    for _ in range(int(rng.integers(2, 6))):
        pos = [rng.uniform(-2.5, 2.5), rng.uniform(-2.5, 2.5), rng.uniform(0.3, 2.0)]
        what = rng.integers(0, 9)
        textured = what <= 2 and rng.integers(3) == 0
        m = _material(b, rng, textured)
        emission = b.diffuse_emission(b.illum(*rng.uniform(1, 4, 3))) if what in (2, 4, 5) and rng.integers(5) == 0 else None   # a second, small light
        if what <= 2:
            P = np.asarray(BOX_P, dtype=np.float32)
            b.add_mesh(BOX_P, BOX_F, m, transform=_xf(rng, pos, rng.uniform(0.2, 0.8, 3)), emission=emission,
                       normals=(P / np.linalg.norm(P, axis=1, keepdims=True)).tolist() if rng.integers(3) == 0 else None,     # smooth-shaded box
                       uvs=((P[:, :2] + 1) / 2).tolist() if textured else None)
        elif what == 3:   # a handful of loose triangles
            p = rng.uniform(-0.8, 0.8, (12, 3)).astype(np.float32)
            b.add_mesh(p.tolist(), [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]], m, transform=_xf(rng, pos, 1.0))
        elif what == 4:
            b.add_sphere(m, radius=float(rng.uniform(0.3, 0.8)), transform=_xf(rng, pos, 1.0), emission=emission)
        elif what == 5:
            b.add_plane(m, width=float(rng.uniform(0.5, 2)), height=float(rng.uniform(0.5, 2)), centering=True, transform=_xf(rng, pos, 1.0), emission=emission)
        elif what == 7:   # the same box twice, different materials: every hit on it is an exact tie, the smaller triangle index must win in both trees
            T = _xf(rng, pos, rng.uniform(0.3, 0.7, 3))
            b.add_mesh(BOX_P, BOX_F, m, transform=T)
            b.add_mesh(BOX_P, BOX_F, _material(b, rng), transform=T)
        elif what == 8:   # degenerate and sliver triangles among ordinary ones, far from the origin of their own mesh
            p = (rng.uniform(-0.5, 0.5, (12, 3)) + 40.0).astype(np.float32)
            p[4] = p[3]; p[5] = p[3]                          # a point
            p[7] = (p[6] + p[8]) / 2                          # three collinear vertices
            p[10] = p[9] + np.float32(1e-6)                   # a sliver
            T = _xf(rng, pos, 1.0); T[:3, 3] -= T[:3, :3] @ np.full(3, 40.0, np.float32)
            b.add_mesh(p.tolist(), [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]], m, transform=T)
        else:
            b.add_quadric(m, [1.0, float(rng.uniform(0.5, 2)), float(rng.uniform(0.5, 3)), -0.25], (-0.6, -0.6, -0.6), (0.6, 0.6, 0.6), transform=_xf(rng, pos, 1.0))
    if light == 1:
        b.environment_light(b.smul(b.illuminant_d65(), b.spectrum_const(float(rng.uniform(0.2, 1.0)))))
    elif light == 2:
        b.distant_light(b.illum(2, 2, 2), direction=tuple(rng.uniform(-1, 1, 2)) + (-1.0,))
    elif light == 3:
        b.cie_sky_light(b.illum(1, 1, 1.2), cloudy=bool(rng.integers(2)))
    elif light == 5:
        b.sun_light((2.0e3 * (0.6 + 0.4 * np.sin(np.arange(64) * 0.11 + rng.uniform(0, 3)))).astype(np.float32), float(rng.uniform(0.3, 1.4)), float(rng.uniform(0, 6)), radius=float(rng.choice([0.5, 1.0, 6.0])))
    elif light == 6:   # textured environment: a random low-resolution map, importance sampled or not
        img = rng.uniform(0.0, 1.0, (int(rng.integers(2, 6)), int(rng.integers(2, 9)), 3)) ** 3
        b.environment_light(b.illuminant_d65(), image=b.rgb_image_to_coefficients(img.astype(np.float32)), distribution=bool(rng.integers(3)), compensation=bool(rng.integers(4) == 0))
    elif light == 4:
        b.sky_light(scene.hosek_sky_table(float(rng.uniform(0.2, 1.3)), float(rng.uniform(0, 6)), turbidity=float(rng.uniform(2, 6)), elevation_count=16, azimuth_count=32), extend=bool(rng.integers(2)))
    eye = np.array([rng.uniform(-1, 1), -7.0 + rng.uniform(-1, 1), rng.uniform(1.0, 3.0)])
    fwd = np.array([0.0, 0.0, 1.0]) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    cam = np.eye(4, dtype=np.float32)
    cam[:3, 0], cam[:3, 1], cam[:3, 2], cam[:3, 3] = right, up, fwd, eye
    kind = rng.integers(0, 5)
    if kind <= 1:
        b.set_camera(cam, width=1.2, height=1.2 * height / width)
    elif kind == 2:
        b.set_camera(cam, width=8.0, height=8.0 * height / width, ortho=True)
    elif kind == 3:
        b.set_spherical_camera(cam)
    else:
        b.set_fisheye_camera(cam)
    return b.build(), multi_tap


SOAK_ITERS = int(os.environ.get("PRGPU_TEST_RANDOM_ITERS", "4"))                      # soak runs: more samples per pixel (the sampler schedule is 4: set both) ...
SOAK_FILM = tuple(int(v) for v in os.environ.get("PRGPU_TEST_RANDOM_FILM", "56x40").split("x"))   # ... and a larger film (with PRGPU_PP_MAX_BLOCKS: more pixels than path slots)
N_SCENES = int(os.environ.get("PRGPU_TEST_RANDOM_SCENES", "40"))   # more for a soak run: 800 scenes pass (round 3)


@pytest.mark.parametrize("seed", list(range(1, N_SCENES + 1)))
def test_random_scene_matches_the_checker(seed):
    sc, multi_tap = random_scene(seed, *SOAK_FILM)
    g = backend.RenderContext(sc)
    o = ob.OracleScene(sc)
    if seed % 5 == 0:   # a rank's tile share instead of the whole film
        from pearray_amd import tiling
        tiles = tiling.tiles_for_rank(sc.width, sc.height, seed % 3, 3, tile=8)
        g.setTiles(tiles); o.set_tiles(tiles)
    g.render(1); g.render(SOAK_ITERS - 1)
    g.waitForFinish()
    o.render(SOAK_ITERS, threads=8)
    gx, gs, gf = g.output(); ox, os_, of = o.output()
    assert np.array_equal(g.primaryHits()[0], o.primary_hits()[0]) and np.array_equal(g.primaryHits()[1], o.primary_hits()[1]), "primary hits"
    assert np.array_equal(gs, os_) and np.array_equal(gf, of), "sample / feedback planes"
    assert g.statistics() == o.statistics()
    assert np.isfinite(gx).all()
    if multi_tap:
        d = np.linalg.norm((gx - ox).ravel()) / max(np.linalg.norm(ox.ravel()), 1e-20)
        assert d <= 1e-5, d
    else:
        assert np.array_equal(gx, ox), float(np.abs(gx - ox).max())


@pytest.mark.parametrize("seed,mode", [(3, "lockstep"), (7, "streaming"), (11, "lockstep"), (19, "streaming"), (23, "lockstep"), (31, "streaming")])
def test_random_scene_in_the_wavefront_pipelines(monkeypatch, seed, mode):
    sc, multi_tap = random_scene(seed)
    ref = backend.RenderContext(sc); ref.render(4); ref.waitForFinish()
    monkeypatch.setenv("PRGPU_MODE", mode)
    g = backend.RenderContext(sc); g.render(2); g.render(2); g.waitForFinish()
    assert g.statistics() == ref.statistics() and np.array_equal(g.output()[1], ref.output()[1])
    assert np.array_equal(g.output()[0], ref.output()[0])


@pytest.mark.parametrize("seed", [2, 5, 9, 14, 20, 27, 33, 38])
def test_random_scene_ray_service(seed):
    """prgpu_trace_closest / prgpu_trace_any over the same random scenes (ties, degenerate triangles, spheres and quadrics included): entity,
    primitive, barycentrics, distance and occlusion equal the checker's tree walk, and the checker's brute-force loop over every primitive."""
    sc, _ = random_scene(seed)
    g, o = backend.RenderContext(sc), ob.OracleScene(sc)
    rng = np.random.default_rng(1000 + seed)
    n = 6000
    org = np.stack([rng.uniform(-4, 4, n), rng.uniform(-7, 4, n), rng.uniform(0.05, 4, n)], 1).astype(np.float32)
    d = np.stack([rng.uniform(-3, 3, n), rng.uniform(-3, 3, n), rng.uniform(0, 2.5, n)], 1).astype(np.float32) - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[::11] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, len(d[::11]))] * rng.choice([-1.0, 1.0], (len(d[::11]), 1)).astype(np.float32)   # axis-parallel rays
    tmin, tmax = np.full(n, 1e-4, np.float32), np.full(n, np.inf, np.float32)
    tmax[::5] = rng.uniform(0.5, 5, len(tmax[::5])).astype(np.float32)
    got = g.traceRays(org, d, tmin, tmax)
    for brute in (False, True):
        want = o.trace_closest(org, d, tmin, tmax, brute=brute)
        for a, b, what in zip(got, want, ("entity", "primitive", "u", "v", "t")):
            assert np.array_equal(a, b), (what, brute, int((a != b).sum()))
    dist = rng.uniform(0.3, 9, n).astype(np.float32)
    occ = g.traceShadowRays(org, d, tmin, dist)
    assert np.array_equal(occ, o.trace_any(org, d, tmin, dist)) and np.array_equal(occ, o.trace_any(org, d, tmin, dist, brute=True))
    assert 0.05 < occ.mean() < 0.999


LPES = ["CE", "CDE", "C.*L", "C<T,S>+.*L", "C[DS]*<R,S>[DS]*E", "CD+L", "C.*B", "C<R,D>{1,2}E", "CS*DL", "C.+<T.>.*"]


@pytest.mark.parametrize("seed", [4, 6, 12, 13, 16, 21, 24, 29, 35, 36, 38, 40])
def test_random_scene_output_planes(seed):
    """The output device over the same random scenes: shading-point AOV planes, the online mean / variance estimator and the planes of four
    randomly drawn light path expressions equal the checker's (bit for bit with a single-tap pixel filter, 1e-5 with a multi-tap one)."""
    sc, multi_tap = random_scene(seed)
    rng = np.random.default_rng(5000 + seed)
    exprs = [str(e) for e in rng.choice(LPES, 4, replace=False)]
    aovs = [str(a) for a in rng.choice(abi.AOV_NAMES, 4, replace=False)]
    g, o = backend.RenderContext(sc), ob.OracleScene(sc)
    for x in (g, o):
        (x.enableAOVs if x is g else x.enable_aovs)(aovs)
        (x.enableVariance if x is g else x.enable_variance)()
        (x.enableLPE if x is g else x.enable_lpe)(exprs)
    g.render(2); g.render(2); g.waitForFinish()
    o.render(4, threads=8)
    same = (lambda a, b: np.array_equal(a, b)) if not multi_tap else (lambda a, b: np.linalg.norm((a - b).ravel()) <= 1e-5 * max(np.linalg.norm(b.ravel()), 1e-20))
    assert same(g.output()[0], o.output()[0]) and g.statistics() == o.statistics()
    for k, e in enumerate(exprs):
        assert same(g.lpe(k), o.lpe(k)), e
    for a in aovs:
        assert np.array_equal(g.aov(a), o.aov(a)), a          # (AOV sums are per pixel, no filter taps)
    gm, gv = g.variance(); om, ov = o.variance()
    assert same(gm, om) and same(gv, ov)
