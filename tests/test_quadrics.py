"""Quadric entities (`quadric`, `cone`, `cylinder`; src/plugins/main/entities/quadric.cpp, src/core/geometry/Quadric.h): the reference's own
known answers (src/tests/quadric.cpp), closed forms, the loader's parametrisations and the callbacks' quirks on the CPU; the HIP path
against the checker on the GPU."""
import ctypes as C

import numpy as np
import pytest

from pearray_amd import _cabi as abi
from pearray_amd import scene, tiling
import oracle_binding as ob
from oracle_binding import f32

SPHERE = [1, 1, 1, 0, 0, 0, 0, 0, 0, -1]      # src/tests/quadric.cpp:9-11
CONE = [1, 1, -1, 0, 0, 0, 0, 0, 0, 0]
CYLINDER = [1, 1, 0, 0, 0, 0, 0, 0, 0, -1]


def intersect(q, o, d):
    t = C.c_float()
    hit = ob.load().orc_quadric_intersect(f32(*q), f32(*o), f32(*d), C.byref(t))
    return bool(hit), t.value


def normal(q, x):
    n = f32(0, 0, 0)
    ob.load().orc_quadric_normal(f32(*q), f32(*x), n)
    return np.array(list(n))


def test_reference_known_answers():
    """src/tests/quadric.cpp 'Normal sphere' (:24-33) and 'Intersection sphere' (:35-44); 'Eval sphere' through the same polynomial."""
    for axis in range(3):
        e = np.eye(3)[axis]
        assert np.allclose(normal(SPHERE, e), e, atol=1e-6)
    hit, t = intersect(SPHERE, (0, 0, -2), (0, 0, 1))
    assert hit and abs(t - 1) < 1e-5
    hit, _ = intersect(SPHERE, (0, 0, -2), (0, 1, 0))
    assert not hit


def test_roots_follow_the_closed_forms():
    rng = np.random.default_rng(3)
    for _ in range(200):
        o = rng.uniform(-3, 3, 3)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        hit, t = intersect(SPHERE, o, d)                               # |o + t d| = 1
        b, c = 2 * o.dot(d), o.dot(o) - 1
        disc = b * b - 4 * c
        roots = sorted(((-b - np.sqrt(disc)) / 2, (-b + np.sqrt(disc)) / 2)) if disc >= 0 else []
        want = next((r for r in roots if r > 1e-6), None)
        if want is None or abs(disc) < 1e-3:
            assert not hit or abs(disc) < 1e-3
        else:
            assert hit and abs(t - want) < 1e-3 * max(1, abs(want))
        hit, t = intersect(CYLINDER, o, d)                             # x^2 + y^2 = 1
        if hit:
            p = o + t * d
            assert abs(p[0] ** 2 + p[1] ** 2 - 1) < 2e-3
        hit, t = intersect(CONE, o, d)                                 # x^2 + y^2 = z^2
        if hit:
            p = o + t * d
            assert abs(p[0] ** 2 + p[1] ** 2 - p[2] ** 2) < 2e-3 * max(1.0, p.dot(p))
    assert intersect([0, 0, 0, 0, 0, 0, 0, 0, 1, -2], (0, 0, 0), (0, 0, 1)) == (True, 2.0)   # a plane z = 2: the linear branch (Quadric.h:58-66)


def quadric_values(s, e):
    return [s.desc.spectral_tables[s.desc.entities[e].params + k] for k in range(16)]


def test_loader_parametrisations_follow_the_reference():
    """QuadricEntityPlugin::create (quadric.cpp:252-316)."""
    src = """(scene :render_width 8 :render_height 8 (camera :name 'c' :type 'standard') (material :name 'm' :type 'diffuse')
      (entity :name 'a' :type 'cylinder' :material 'm' :radius 2 :height 3)
      (entity :name 'b' :type 'cylinder' :material 'm' :radius 2 :height 3 :center_on false)
      (entity :name 'c' :type 'cone' :material 'm' :radius 2 :height 4)
      (entity :name 'd' :type 'cone' :material 'm' :radius 2 :height 4 :center_on false)
      (entity :name 'e' :type 'quadric' :material 'm' :parameters [1,2,3])
      (entity :name 'f' :type 'quadric' :material 'm' :parameters [1,2,3,-4] :min [-2,-3,-4] :max [2,3,4])
      (entity :name 'g' :type 'quadric' :material 'm' :parameters [1,2,3,4,5,6,7,8,9,10]))"""
    s = scene.PrcScene(source=src)
    assert [s.desc.entities[i].kind for i in range(7)] == [abi.ENTITY_QUADRIC] * 7 and s.desc.n_triangles == 7
    assert quadric_values(s, 0) == [0.25, 0.25, 0, 0, 0, 0, 0, 0, 0, -1, -2, -2, -1.5, 2, 2, 1.5]
    assert quadric_values(s, 1) == [0.25, 0.25, 0, 0, 0, 0, 0, 0, 0, -1, -2, -2, 0, 2, 2, 3]
    assert quadric_values(s, 2) == [0.25, 0.25, -0.0625, 0, 0, 0, 0, 0, 0.25, -0.25, -2, -2, -2, 2, 2, 2]
    assert quadric_values(s, 3) == [0.25, 0.25, -0.0625, 0, 0, 0, 0, 0, 0, 0, -2, -2, 0, 2, 2, 4]
    assert quadric_values(s, 4) == [1, 2, 3, 0, 0, 0, 0, 0, 0, 0, -1, -1, -1, 1, 1, 1]
    assert quadric_values(s, 5) == [1, 2, 3, 0, 0, 0, 0, 0, 0, -4, -2, -3, -4, 2, 3, 4]
    assert quadric_values(s, 6) == [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, -1, -1, -1, 1, 1, 1]
    for bad, code, needle in (("(entity :type 'quadric' :material 'm' :parameters [1,2])", -1, "quadric parameters"),
                              ("(emission :name 'em' :type 'standard' :radiance 1) (entity :type 'cone' :material 'm' :emission 'em')", -4, "emissive")):
        with pytest.raises(abi.PrgpuError) as e:
            scene.PrcScene(source="(scene (camera :name 'c' :type 'standard') (material :name 'm' :type 'diffuse') %s)" % bad)
        assert needle in str(e.value) and ("error %d" % code) in str(e.value)
    # the builder mirrors the loader
    b = scene.SceneBuilder(8, 8)
    m = b.lambert(b.spectrum_const(0.5))
    for args in (dict(kind="cylinder", radius=2, height=3), dict(kind="cylinder", radius=2, height=3, center_on=False), dict(kind="cone", radius=2, height=4),
                 dict(kind="cone", radius=2, height=4, center_on=False)):
        kind = args.pop("kind")
        (b.add_cylinder if kind == "cylinder" else b.add_cone)(m, **args)
    for k in range(4):
        off = b.entities[k].params
        assert [float(np.float32(v)) for v in b.tables[off:off + 16]] == quadric_values(s, k)


def callback_scene(transform=scene.IDENTITY, params=SPHERE, lo=(-1, -1, -1), hi=(1, 1, 1)):
    b = scene.SceneBuilder(8, 8)
    m = b.lambert(b.spectrum_const(0.5))
    b.add_quadric(m, params, lo, hi, transform=transform)
    return b.build()


def test_callbacks_clip_to_the_box_and_occlusion_does_not():
    """userIntersectFuncN clips the hit to the local box and the ray's extent (quadric.cpp:166-186); userOccludedFuncN tests the unbounded
    surface from the box's entry on (:222-230) -- a cylinder cut to |z| <= 0.5 still shadows along rays that pass its box beside the cut."""
    sc = callback_scene(params=CYLINDER, lo=(-1, -1, -0.5), hi=(1, 1, 0.5))
    o = ob.OracleScene(sc)
    lib = ob.load()
    t = C.c_float()
    assert lib.orc_quadric_closest(o.h, f32(-3, 0, 0), f32(1, 0, 0), 1e-4, np.inf, C.byref(t)) == 0 and abs(t.value - 2) < 1e-5
    assert lib.orc_quadric_closest(o.h, f32(-3, 0, 0), f32(1, 0, 0), 1e-4, 1.5, C.byref(t)) == 0xFFFFFFFF            # beyond the ray's extent
    assert lib.orc_quadric_closest(o.h, f32(-3, 0, 0.8), f32(1, 0, 0), 1e-4, np.inf, C.byref(t)) == 0xFFFFFFFF        # above the cut: misses the box
    # a ray that enters the box through its top inside the cylinder and leaves through the bottom never meets the surface in the box ...
    d = np.array([0.1, 0, -1.0]); d /= np.linalg.norm(d)
    assert lib.orc_quadric_closest(o.h, f32(0, 0, 3), f32(*d), 1e-4, np.inf, C.byref(t)) == 0xFFFFFFFF
    # ... but the infinite cylinder is met further down: the occlusion callback reports it
    assert lib.orc_quadric_occluded(o.h, f32(0, 0, 3), f32(*d), 1e-4, 100.0) == 1
    assert lib.orc_quadric_occluded(o.h, f32(0, 0, 3), f32(0, 0, -1), 1e-4, 100.0) == 0                                 # along the axis: no root at all
    assert lib.orc_quadric_occluded(o.h, f32(5, 5, 3), f32(0, 0, -1), 1e-4, 100.0) == 0                                 # misses the bounds: never called
    assert lib.orc_quadric_occluded(o.h, f32(0, 0, 3), f32(*d), 1e-4, 2.0) == 0                                         # the extent ends before the bounds


def test_the_local_parameter_is_the_world_distance_under_affine_maps():
    """quadric.cpp:185 stores the LOCAL parameter; with an unnormalised local direction that is the world distance (affine maps keep the
    parameter of a point on a line), whatever the entity's scale."""
    T = np.diag([2.0, 0.5, 3.0, 1.0]); T[:3, 3] = [1, -2, 0.5]
    o = ob.OracleScene(callback_scene(transform=T))                     # an ellipsoid with half axes 2, 0.5, 3 around (1, -2, 0.5)
    t = C.c_float()
    lib = ob.load()
    assert lib.orc_quadric_closest(o.h, f32(-4, -2, 0.5), f32(1, 0, 0), 1e-4, np.inf, C.byref(t)) == 0 and abs(t.value - 3) < 1e-5
    assert lib.orc_quadric_closest(o.h, f32(1, -2, 10), f32(0, 0, -1), 1e-4, np.inf, C.byref(t)) == 0 and abs(t.value - 6.5) < 1e-5


def test_quadric_sphere_renders_like_the_analytic_sphere():
    """x^2 + y^2 + z^2 = 1 as a quadric entity and as a sphere entity: the same hits (entity ids) and the same image up to the last bits
    of two different intersection formulas."""
    def build(kind):
        b = scene.SceneBuilder(48, 36)
        b.settings.aa_sampler, b.settings.aa_samples, b.settings.filter, b.settings.filter_radius = abi.SAMPLER_SOBOL, 16, abi.FILTER_BLOCK, 0
        grey = b.lambert(b.spectrum_const(0.6))
        b.add_mesh([[-4, -1, -4], [4, -1, -4], [4, -1, 4], [-4, -1, 4]], [[0, 1, 2], [0, 2, 3]], grey)
        if kind == "sphere":
            b.add_sphere(grey, 1.0)
        else:
            b.add_quadric(grey, SPHERE, (-1, -1, -1), (1, 1, 1))
        b.environment_light(b.spectrum_const(1.0))
        cam = np.eye(4, dtype=np.float32); cam[:3, 3] = [0, 0.5, 5]
        b.set_camera(cam, local_direction=(0, 0, -1))
        return b.build()
    a, q = build("sphere"), build("quadric")
    oa, oq = ob.OracleScene(a), ob.OracleScene(q)
    oa.render(16, threads=8); oq.render(16, threads=8)
    ea, _ = oa.primary_hits(); eq, _ = oq.primary_hits()
    assert (ea != eq).mean() < 0.01
    xa, xq = oa.output()[0], oq.output()[0]
    assert abs(xa.mean() - xq.mean()) < 0.01 * xa.mean()


QUADRIC_SCENE = """(scene :render_width %d :render_height %d :camera 'c'
  (sampler :slot 'aa' :type 'sobol' :sample_count %d)
  (filter :slot 'pixel' :type 'block' :radius 0)
  (camera :name 'c' :type 'standard' :width 1 :height 0.75 :local_direction [0,0,-1] :local_up [0,1,0] :local_right [1,0,0] :position [0,1.2,5])
  (light :name 'env' :type 'env' :radiance (illuminant 'D65'))
  (emission :name 'lamp' :type 'standard' :radiance (illum 8 8 7))
  (material :name 'white' :type 'diffuse' :albedo (refl 0.7 0.7 0.7))
  (material :name 'red' :type 'diffuse' :albedo (refl 0.8 0.2 0.2))
  (material :name 'glass' :type 'glass' :index 1.5)
  (material :name 'metal' :type 'conductor' :eta 0.2 :k 3.9 :roughness 0.2)
  (mesh :name 'quad' (attribute :type 'p' [-1,0,-1],[1,0,-1],[1,0,1],[-1,0,1]) (faces [0,1,2,3]))
  (entity :name 'floor' :type 'mesh' :mesh 'quad' :materials 'white' :scale 4)
  (entity :name 'lamp' :type 'mesh' :mesh 'quad' :materials 'white' :emission 'lamp' :rotation (euler 180 0 0) :position [0,3,1] :scale 0.5)
  (entity :name 'cone' :type 'cone' :material 'red' :radius 0.6 :height 1.5 :center_on false :rotation (euler -90 0 0) :position [-1.4,0,0])
  (entity :name 'cyl' :type 'cylinder' :material 'metal' :radius 0.5 :height 1.2 :rotation (euler -90 0 0) :position [1.4,0.6,0] :scale [1,0.7,1])
  (entity :name 'ell' :type 'quadric' :material 'glass' :parameters [1,2,4,-0.25] :min [-0.6,-0.6,-0.6] :max [0.6,0.6,0.6] :position [0,0.6,0.5])
  (entity :name 'hyp' :type 'quadric' :material 'white' :parameters [4,4,-1,0,0,0,0,0,0,-0.04] :min [-0.4,-0.4,-0.5] :max [0.4,0.4,0.5] :rotation (euler -90 0 0) :position [0,0.5,-1.5]))"""


def test_checker_renders_the_quadric_scene():
    s = scene.PrcScene(source=QUADRIC_SCENE % (64, 48, 8))
    o = ob.OracleScene(s)
    o.render(8, threads=8)
    xyz, smp, fb = o.output()
    ent, _ = o.primary_hits()
    assert np.isfinite(xyz).all() and not fb.any()
    seen = set(np.unique(ent).tolist())
    assert {0, 2, 3, 4, 5} <= seen                                     # floor, cone, cylinder, ellipsoid, hyperboloid (the lamp faces down)
    o2 = ob.OracleScene(s); o2.render(8, threads=3)
    assert np.array_equal(o2.output()[0], xyz)


@pytest.mark.gpu
def test_gpu_quadric_scene_is_bit_exact_and_shards_over_tiles():
    from pearray_amd import backend
    s = scene.PrcScene(source=QUADRIC_SCENE % (96, 72, 8))
    g = backend.RenderContext(s)
    for n in (3, 5):
        g.render(n)
    g.waitForFinish()
    o = ob.OracleScene(s); o.render(8, threads=16)
    gx, gs, gf = g.output(); ox, os_, of = o.output()
    ge, gp = g.primaryHits(); oe, op = o.primary_hits()
    assert np.array_equal(ge, oe) and np.array_equal(gp, op)
    assert np.array_equal(gs, os_) and np.array_equal(gf, of) and g.statistics() == o.statistics()
    assert np.array_equal(gx, ox)
    acc = np.zeros_like(gx)
    for rank in range(2):
        h = backend.RenderContext(s); h.setTiles(tiling.tiles_for_rank(96, 72, rank, 2, tile=16)); h.render(8); h.waitForFinish()
        acc += h.output()[0]
    assert np.array_equal(acc, gx)
    # light path expressions live in the same kernel variant: planes of paths over quadric surfaces
    exprs = ["CDE", "C<T.>+.*L", "C.*<RS>.*L"]
    gl = backend.RenderContext(s); gl.enableLPE(exprs); gl.render(8); gl.waitForFinish()
    ol = ob.OracleScene(s); ol.enable_lpe(exprs); ol.render(8, threads=16)
    assert np.array_equal(gl.output()[0], gx)
    for k in range(len(exprs)):
        assert np.array_equal(gl.lpe(k), ol.lpe(k)), exprs[k]
    assert gl.lpe(1).any() and gl.lpe(2).any()                       # through the glass ellipsoid, off the metal cylinder


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["lockstep", "streaming"])
def test_gpu_quadrics_in_the_wavefront_pipelines(monkeypatch, mode):
    """The quadric test sits where a lane picks a ray up, in every tracing kernel: the lockstep and streaming pipelines render the scene
    like the persistent kernel and the checker do, bit for bit."""
    from pearray_amd import backend
    monkeypatch.setenv("PRGPU_MODE", mode)
    s = scene.PrcScene(source=QUADRIC_SCENE % (96, 72, 6))
    g = backend.RenderContext(s); g.render(2); g.render(4); g.waitForFinish()
    o = ob.OracleScene(s); o.render(6, threads=16)
    gx, gs, gf = g.output(); ox, os_, of = o.output()
    assert np.array_equal(g.primaryHits()[0], o.primary_hits()[0])
    assert np.array_equal(gs, os_) and np.array_equal(gf, of) and g.statistics() == o.statistics()
    assert np.array_equal(gx, ox)


@pytest.mark.gpu
def test_gpu_ray_service_traces_quadrics():
    """prgpu_trace_closest / prgpu_trace_any (the IArchive surface) over a scene with quadric entities: entity, primitive, distance and
    occlusion equal the checker's for rays through the cone, the cylinder, the ellipsoid, the hyperboloid and past them -- including the
    reference's occlusion callback, which reports the unbounded surface behind a quadric's box."""
    from pearray_amd import backend
    s = scene.PrcScene(source=QUADRIC_SCENE % (32, 24, 1))
    g, o = backend.RenderContext(s), ob.OracleScene(s)
    rng = np.random.default_rng(11)
    n = 20000
    org = np.stack([rng.uniform(-3, 3, n), rng.uniform(0.05, 3, n), rng.uniform(-3, 5, n)], 1).astype(np.float32)
    target = np.stack([rng.uniform(-2, 2, n), rng.uniform(0, 1.5, n), rng.uniform(-2, 1, n)], 1).astype(np.float32)
    d = target - org; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tmin, tmax = np.full(n, 1e-4, np.float32), np.full(n, np.inf, np.float32)
    tmax[::7] = 2.5                                                      # some rays end before they reach anything
    ge, gp, gu, gv, gt = g.traceRays(org, d, tmin, tmax)
    oe, op, ou, ov, ot = o.trace_closest(org, d, tmin, tmax)
    assert np.array_equal(ge, oe) and np.array_equal(gp, op) and np.array_equal(gt, ot) and np.array_equal(gu, ou) and np.array_equal(gv, ov)
    hit_kinds = set(np.unique(ge).tolist())
    assert {0, 2, 3, 4, 5} <= hit_kinds                                  # floor, cone, cylinder, ellipsoid, hyperboloid
    dist = rng.uniform(0.5, 8, n).astype(np.float32)
    assert np.array_equal(g.traceShadowRays(org, d, tmin, dist), o.trace_any(org, d, tmin, dist))
