"""End-to-end checks of the CPU oracle itself: BVH vs brute force, an analytic radiometric known answer
(the scene of src/tests/python/validity.py:10-108), sampler/table properties, tile independence."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from pearray_amd import _cabi as abi
from pearray_amd import scene, tiling


def _random_rays(n, seed, lo=(-1, -1, 0), hi=(1, 1, 2)):
    rng = np.random.default_rng(seed)
    org = (rng.random((n, 3)) * (np.array(hi) - lo) + lo).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return org, d


def test_bvh_matches_brute_force_cornell():
    o = ob.OracleScene(scene.cornell_box(16, 16, spp=1))
    org, d = _random_rays(4000, 1)
    a = o.trace_closest(org, d, 1e-4, np.inf)
    b = o.trace_closest(org, d, 1e-4, np.inf, brute=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert (a[0] != abi.INVALID_ID).mean() > 0.8
    dist = np.where(a[0] != abi.INVALID_ID, a[4], 10.0).astype(np.float32)
    assert np.array_equal(o.trace_any(org, d, 1e-4, dist * 2), o.trace_any(org, d, 1e-4, dist * 2, brute=True))
    # a shadow ray that stops short of the closest hit is never occluded (Scene.cpp:275: tfar = distance - 0.001)
    assert not o.trace_any(org, d, 1e-4, dist * 0.5, brute=True)[a[0] != abi.INVALID_ID].any()


def test_bvh_matches_brute_force_soup():
    o = ob.OracleScene(scene.cornell_soup(16, 16, spp=1, n_triangles=5032))
    org, d = _random_rays(1500, 2, lo=(-0.9, -0.9, 0.1), hi=(0.9, 0.9, 1.8))
    a = o.trace_closest(org, d, 1e-4, np.inf)
    b = o.trace_closest(org, d, 1e-4, np.inf, brute=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_hit_barycentrics_reconstruct_point():
    sc = scene.cornell_box(16, 16, spp=1)
    o = ob.OracleScene(sc)
    org, d = _random_rays(500, 3)
    ent, prim, u, v, t = o.trace_closest(org, d, 1e-4, np.inf)
    for i in np.nonzero(ent != abi.INVALID_ID)[0][:100]:
        e = sc.entities[int(ent[i])]
        tri = sc.indices[e.first_tri + int(prim[i])]
        p0, p1, p2 = sc.positions[tri]  # Cornell entity transforms are the identity
        p = (1 - u[i] - v[i]) * p0 + u[i] * p1 + v[i] * p2  # Triangle.h:22-27
        assert np.allclose(p, org[i] + t[i] * d[i], atol=2e-5)


def _form_factor(p):
    """analytical() of validity.py:21-38: point-to-rectangle form factor of the 1x1 light at z = 2."""
    Rx, Ry, Ro, Np = np.array([1, 0, 0.]), np.array([0, 1, 0.]), np.array([-0.5, -0.5, 2.]), np.array([0, 0, 1.])
    n = lambda k: k / np.linalg.norm(k)
    R0 = Ro - [p[0], p[1], 0]
    R = [R0, R0 + Rx, R0 + Rx + Ry, R0 + Ry]
    K = sum(np.arccos(np.dot(n(R[i]), n(R[(i + 1) % 4]))) * n(np.cross(R[i], R[(i + 1) % 4])) for i in range(4))
    return abs(0.5 / np.pi * np.dot(Np, K))


def test_analytic_form_factor():
    """Unit diffuse plane (albedo 1) under a 1x1 unit-radiance area light at z=2, direct lighting only
    (NEE, max_ray_depth 1, no MIS partner): reflected radiance = form factor, so pixel Y = F(p)."""
    W = 40
    b = scene.SceneBuilder(W, W)
    s = b.settings
    s.aa_sampler, s.aa_samples, s.max_ray_depth, s.direct, s.filter, s.filter_radius = abi.SAMPLER_MJITT, 256, 1, 0, abi.FILTER_BLOCK, 0
    white = b.lambert(b.spectrum_const(1.0))
    ems = b.diffuse_emission(b.spectrum_const(1.0))
    # vertex normals given: without them the reference builds the tangent frame from the (non-orthogonal)
    # triangle edges dPdu/dPdv (mesh.cpp:216-219), which skews cos(theta) -- a reference quirk kept by the
    # oracle but unwanted in a radiometric known-answer test
    b.add_mesh([[-2, -2, 0], [2, -2, 0], [2, 2, 0], [-2, 2, 0]], [[0, 1, 2], [0, 2, 3]], white, normals=[[0, 0, 1]] * 4)
    b.add_mesh([[-0.5, -0.5, 2], [-0.5, 0.5, 2], [0.5, 0.5, 2], [0.5, -0.5, 2]], [[0, 1, 2], [0, 2, 3]], white, emission=ems,
               normals=[[0, 0, -1]] * 4)
    T = np.eye(4, dtype=np.float32); T[2, 3] = 1.0
    b.set_camera(T, width=2.0, height=2.0, local_direction=(0, 0, -1), local_right=(1, 0, 0), local_up=(0, 1, 0))
    o = ob.OracleScene(b.build())
    o.render(256, threads=8)
    xyz, smp, fb = o.output()
    assert (fb == 0).all() and (smp == 256).all()
    for fx, fy in ((0.5, 0.5), (0.25, 0.25), (0.75, 0.25), (0.25, 0.75), (0.75, 0.75)):
        px, py = int(W * fx), int(W * fy)
        # pixel p covers [p-0.5, p+0.5) (RenderTile.cpp:85: Pixel = p + aa - 0.5); camera at z=1 looking down,
        # sensor 2x2 => plane point x = nx, y = -ny
        x, y = 2 * (px / W - 0.5), -2 * (py / W - 0.5)
        block = xyz[py - 1:py + 2, px - 1:px + 2, 1].mean()
        assert abs(block - _form_factor((x, y))) < 0.02 * _form_factor((x, y)) + 1e-3, (fx, fy, block, _form_factor((x, y)))


def test_sobol_table_is_a_shuffled_01_sequence():
    o = ob.OracleScene(scene.cornell_box(8, 8, spp=64, sampler=abi.SAMPLER_SOBOL))
    n, ptr = C.c_uint32(), C.POINTER(C.c_float)()
    o.lib.orc_sobol_table(o.h, C.byref(n), C.byref(ptr))
    t = np.ctypeslib.as_array(ptr, shape=(2 * n.value,)).reshape(-1, 2).copy()
    assert n.value == 64
    # (0,2)-sequence property: the first 64 points hit every 8x8 cell exactly once
    cells = (np.floor(t[:, 0] * 8) * 8 + np.floor(t[:, 1] * 8)).astype(int)
    assert sorted(cells) == list(range(64))
    assert sorted(np.round(t[:, 0] * 64).astype(int)) == list(range(64))  # dimension 0 = van der Corput


def test_mjitt_samples_are_stratified():
    o = ob.OracleScene(scene.cornell_box(8, 8, spp=16, sampler=abi.SAMPLER_MJITT))
    s = C.c_uint64()
    o.lib.orc_pcg_seed(1, C.byref(s))
    pts = []
    for i in range(16):
        out = ob.f32(0, 0)
        o.lib.orc_sampler_2d(o.h, C.byref(s), i, out)
        pts.append(out[:])
    pts = np.array(pts)
    assert sorted(np.floor(pts[:, 1] * 16).astype(int)) == list(range(16))  # PR_MJS_CLIP: y stratified in n
    assert (pts >= 0).all() and (pts < 1).all()


def test_wavelength_cdf_and_light_selector():
    o = ob.OracleScene(scene.cornell_box(8, 8, spp=1))
    n, ptr = C.c_uint32(), C.POINTER(C.c_float)()
    o.lib.orc_wavelength_cdf(o.h, C.byref(n), C.byref(ptr))
    cdf = np.ctypeslib.as_array(ptr, shape=(n.value,))
    assert n.value == 441 and cdf[0] == 0 and cdf[-1] == 1 and (np.diff(cdf) > 0).all()
    nl, c, i = C.c_uint32(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    o.lib.orc_light_selector(o.h, C.byref(nl), C.byref(c), C.byref(i))
    assert nl.value == 1 and c[0] == 0 and c[1] == 1 and abs(i[0] - 1.0) < 1e-6
    assert o.lib.orc_rr_probability(o.h, 4) == 1.0 and abs(o.lib.orc_rr_probability(o.h, 5) - 0.9) < 1e-7  # RussianRoulette.h:22-35


def test_rank_sharded_render_sums_to_whole():
    """Tile ownership (multi-GPU model): per-rank frames of disjoint tiles add up to the full render."""
    W, H, spp = 48, 40, 4
    whole = ob.OracleScene(scene.cornell_box(W, H, spp=spp))
    whole.render(spp, threads=4)
    ref_xyz, ref_smp, _ = whole.output()
    acc, acc_smp = np.zeros_like(ref_xyz), np.zeros_like(ref_smp)
    for rank in range(3):
        o = ob.OracleScene(scene.cornell_box(W, H, spp=spp))
        o.set_tiles(tiling.tiles_for_rank(W, H, rank, 3, tile=16))
        o.render(spp, threads=4)
        x, s, _ = o.output()
        acc += x
        acc_smp += s
    assert np.array_equal(acc_smp, ref_smp)
    assert np.array_equal(acc, ref_xyz)  # delta filter: disjoint pixels, adding zeros is exact


def test_render_is_deterministic_across_thread_counts():
    a = ob.OracleScene(scene.cornell_box(32, 32, spp=2)); a.render(2, threads=1)
    b = ob.OracleScene(scene.cornell_box(32, 32, spp=2)); b.render(2, threads=8)
    assert np.array_equal(a.output()[0], b.output()[0]) and a.statistics() == b.statistics()


def test_oracle_rejects_malformed_scenes():
    sc = scene.cornell_box(8, 8, spp=1)
    sc.desc.api_version = 99
    assert ob.load().orc_scene_create(C.byref(sc.desc)) is None
    sc = scene.cornell_box(8, 8, spp=1)
    sc.indices[0] = 10_000
    assert ob.load().orc_scene_create(C.byref(sc.desc)) is None


def test_halton_radical_inverse_known_values():
    lib = ob.load()
    # https://en.wikipedia.org/wiki/Halton_sequence (cited by HaltonSampler.cpp:25-27): base 2 and base 3 prefixes
    assert [lib.orc_halton(i, 2) for i in range(1, 9)] == [0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875, 0.0625]
    want3 = [1 / 3, 2 / 3, 1 / 9, 4 / 9, 7 / 9, 2 / 9, 5 / 9, 8 / 9, 1 / 27]
    assert np.allclose([lib.orc_halton(i, 3) for i in range(1, 10)], want3, atol=1e-7)
    assert lib.orc_halton(0, 13) == 0.0


@pytest.mark.parametrize("kind", [abi.SAMPLER_HALTON, abi.SAMPLER_HAMMERSLEY])
def test_halton_and_hammersley_tables(kind):
    sc = scene.cornell_box(8, 8, spp=32, sampler=kind)
    o = ob.OracleScene(sc)
    n, ptr = C.c_uint32(), C.POINTER(C.c_float)()
    o.lib.orc_sobol_table(o.h, C.byref(n), C.byref(ptr))
    t = np.ctypeslib.as_array(ptr, shape=(2 * n.value,)).reshape(-1, 2).copy()
    lib = ob.load()
    burnin = 47 if kind == abi.SAMPLER_HALTON else 13     # defaults: max(13, 47) resp. base_x (HaltonSampler.cpp:173,191)
    assert np.array_equal(t[:, 0], np.array([lib.orc_halton(i + burnin, 13) for i in range(32)], dtype=np.float32))
    if kind == abi.SAMPLER_HALTON:
        assert np.array_equal(t[:, 1], np.array([lib.orc_halton(i + burnin, 47) for i in range(32)], dtype=np.float32))
    else:
        assert np.array_equal(t[:, 1], ((np.float32(0.5) + np.arange(32, dtype=np.float32)) / np.float32(32)))
    assert (t >= 0).all() and (t < 1).all()
    o.render(3, threads=2)
    assert np.isfinite(o.output()[0]).all()


def test_lanczos_filter_weights_and_stratified_sampler():
    """LanczosFilter.cpp:38-66: windowed sinc, normalised over the mirrored quadrant; StratifiedSampler.cpp: jittered grid."""
    sc = scene.cornell_box(24, 24, spp=16, sampler=abi.SAMPLER_STRATIFIED, filter=abi.FILTER_LANCZOS, filter_radius=3)
    o = ob.OracleScene(sc)
    o.render(16, threads=2)
    xyz, smp, fb = o.output()
    assert np.isfinite(xyz).all() and smp.max() == 16
    ref = ob.OracleScene(scene.cornell_box(24, 24, spp=16, sampler=abi.SAMPLER_STRATIFIED)); ref.render(16, threads=2)
    # commitSpectrals2 only splats taps with weight > eps: the negative lobes of the normalised kernel are dropped, the positive
    # ones alone sum to more than one
    assert 1.0 < xyz.sum() / ref.output()[0].sum() < 3.0
    u = ob.OracleScene(scene.cornell_box(24, 24, spp=4, sampler=abi.SAMPLER_UNIFORM)); u.render(4, threads=2)
    assert np.isfinite(u.output()[0]).all()


def test_analytic_sphere_hits_match_the_closed_form():
    """sphere.cpp: centre = M * 0, radius = r * mean column norm; hits from the Embree-style projection formula."""
    b = scene.SceneBuilder(8, 8)
    b.set_camera(scene.IDENTITY)
    m = b.lambert(b.spectrum_const(0.5))
    T = np.array([[2, 0, 0, 1.0], [0, 2, 0, -0.5], [0, 0, 2, 3.0], [0, 0, 0, 1]], dtype=np.float32)
    b.add_sphere(m, radius=0.75, transform=T)                         # world radius 1.5 at (1, -0.5, 3)
    b.add_mesh([[-9, -9, 9], [9, -9, 9], [9, 9, 9], [-9, 9, 9]], [[0, 1, 2, 3]], m)
    sc = b.build()
    o = ob.OracleScene(sc)
    rng = np.random.default_rng(5)
    n = 4000
    org = (rng.random((n, 3)) * 8 - 4).astype(np.float32)
    org = org[np.linalg.norm(org - [1, -0.5, 3], axis=1) > 1.6]      # outside the sphere
    d = rng.normal(size=(len(org), 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ent, prim, u, v, t = o.trace_closest(org, d, 1e-4, np.inf)
    eb, pb, ub, vb, tb = o.trace_closest(org, d, 1e-4, np.inf, brute=True)
    assert np.array_equal(ent, eb) and np.array_equal(t, tb)          # BVH == brute force
    c = np.array([1, -0.5, 3], dtype=np.float64)
    oc = org.astype(np.float64) - c
    bq = (oc * d).sum(1); cq = (oc * oc).sum(1) - 1.5 ** 2
    disc = bq * bq - cq
    t_ref = np.where(disc >= 0, -bq - np.sqrt(np.maximum(disc, 0)), np.inf)
    hits = (ent == 0)
    assert hits.sum() > 100 and (prim[hits] == 0).all() and (u[hits] == 0).all()
    assert np.allclose(t[hits], t_ref[hits], rtol=2e-5, atol=2e-5)
    front = (disc > 1e-4) & (t_ref > 1e-3)
    assert (ent[front & (t_ref < tb + 1e-3)] == 0).all()
    # from inside: the back root
    inside = np.tile(np.array([[1, -0.5, 3]], dtype=np.float32), (64, 1))
    e2, p2, _, _, t2 = o.trace_closest(inside, d[:64], 1e-4, np.inf)
    assert (e2 == 0).all() and np.allclose(t2, 1.5, atol=1e-5)


def test_worker_tile_grid_and_native_build_do_not_change_the_frame(tmp_path):
    """bench.py's cpu_baseline leg times the checker with 32 x 32-pixel worker tiles and an -O3 -march=native rebuild: both must
    produce the very frame of the default 8 x 8 grid / -O2 build (single-tap filter: tile-layout independent)."""
    import bench
    sc = scene.cornell_box(48, 40, spp=4)
    a = ob.OracleScene(sc)
    a.render(3, threads=2)
    path, how = bench.native_oracle()
    assert path is not None and "-O3" in how
    b = ob.OracleScene(sc, lib=ob.load_from(path))
    b.set_tile_grid(7, 5)
    b.render(3, threads=3)
    for x, y in zip(a.output(), b.output()):
        assert np.array_equal(x, y)
    assert a.statistics() == b.statistics()
    assert b.lib.orc_set_tile_grid(b.h, 0, 4) != 0
