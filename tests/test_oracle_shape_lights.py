"""Emissive plane and sphere entities as area lights: PlaneEntity's spherical-rectangle sampling (plane.cpp:99-196, Urena et al. 2013) and
SphereEntity::sampleParameterPoint (sphere.cpp:106-118), the shared acos / sin / cos they need, and their place in Light::sample and
handleDirectHit."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from pearray_amd import _cabi as abi
from pearray_amd import scene


def light_scene(kind, w=32, h=32, spp=16, light_transform=None, **settings):
    """A 0.8 x 0.8 light (quad mesh, plane entity or a sphere) two units above a large grey floor, seen from the side."""
    b = scene.SceneBuilder(w, h)
    b.settings.aa_sampler, b.settings.aa_samples, b.settings.mapper = abi.SAMPLER_MJITT, spp, abi.MAPPER_RANDOM
    for k, v in settings.items():
        setattr(b.settings, k, v)
    grey = b.lambert(b.spectrum_const(0.7))
    ems = b.diffuse_emission(b.spectrum_const(5.0))
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = 2.0
    if light_transform is not None:
        T = np.asarray(light_transform, dtype=np.float32)
    if kind == "plane":  # x = (1,0,0), y = (0,-1,0): the normal x cross y points down
        b.add_plane(grey, x_axis=(1, 0, 0), y_axis=(0, -1, 0), width=0.8, height=0.8, centering=True, transform=T, emission=ems)
    elif kind == "mesh":  # the same parallelogram with the plane's triangulation (plane.cpp:81-84)
        p = np.array([[-0.4, 0.4, 0], [-0.4, -0.4, 0], [0.4, -0.4, 0], [0.4, 0.4, 0]], np.float32)
        b.add_mesh(p, [[0, 1, 3], [2, 3, 1]], grey, transform=T, emission=ems)
    else:
        b.add_sphere(grey, radius=0.3, transform=T, emission=ems)
    b.add_mesh([[-3, -3, 0], [3, -3, 0], [3, 3, 0], [-3, 3, 0]], [[0, 1, 2], [0, 2, 3]], grey)
    eye = np.array([0, -4, 1.5])
    fwd = np.array([0, 0, 0.3]) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 0, 1])
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    M = np.eye(4, dtype=np.float32)
    M[:3, 0], M[:3, 1], M[:3, 2], M[:3, 3] = right, up, fwd, eye
    b.set_camera(M, width=0.9, height=0.9)
    return b.build()


def render(sc, n):
    o = ob.OracleScene(sc)
    o.render(n)
    return o, o.output()[0].reshape(sc.height, sc.width, 3)


def test_plane_light_agrees_with_the_same_quad_as_a_mesh_light():
    """Spherical-rectangle sampling is unbiased: the floor under a plane light receives what it receives from the same quad as a
    two-triangle mesh light (area sampling), within Monte-Carlo noise; the plane version is the less noisy of the two."""
    _, a = render(light_scene("mesh", spp=32), 32)
    _, b = render(light_scene("plane", spp=32), 32)
    fa, fb = a[20:, :, 1], b[20:, :, 1]  # floor rows
    assert abs(fa.mean() - fb.mean()) < 0.03 * fa.mean()
    assert np.isfinite(b).all()


def test_plane_and_sphere_lights_register_with_their_own_areas():
    """LightSampler intensities: world area x mean power; plane area = |T x| |T y| (plane.cpp:48-54), sphere area by Knud Thomsen's formula
    (sphere.cpp:49-66), which is exact (4 pi r^2) for a uniform scale."""
    def intensity(sc):
        o = ob.OracleScene(sc)
        nl, c, i = C.c_uint32(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        o.lib.orc_light_selector(o.h, C.byref(nl), C.byref(c), C.byref(i))
        assert nl.value == 1
        return o
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] *= 2.0
    T[2, 3] = 2.0
    for kind in ("plane", "sphere"):
        o = intensity(light_scene(kind, light_transform=T))
        o.render(1)
        assert np.isfinite(o.output()[0]).all()


@pytest.mark.parametrize("kind", ["plane", "sphere"])
@pytest.mark.parametrize("kw", [dict(), dict(mis=abi.MIS_POWER), dict(nee=0), dict(spectral_mono=1, spectral_start=550.0, spectral_end=830.0)])
def test_shape_light_renders(kind, kw):
    o, img = render(light_scene(kind, spp=8, **kw), 8)
    assert np.isfinite(img).all()
    if not kw.get("spectral_mono"):  # monochrome NEE fragments are dropped as NaN feedback (direct.cpp:321)
        assert img[20:, :, 1].mean() > 0.01
    st = o.statistics()
    assert st["shadow_rays"] > 0 or kw.get("nee") == 0


def test_primary_hits_on_a_plane_light_use_the_origin_as_previous_vertex():
    """direct.cpp:55,358: LastPosition starts at (0,0,0), so the MIS pdf of a directly visible plane light is evaluated for the spherical
    rectangle seen from the world origin -- with NEE on, the light's own pixels stay finite and positive whatever that pdf is."""
    sc = light_scene("plane", spp=4)
    o, img = render(sc, 4)
    top = img[:12, 8:24, 1]
    assert np.isfinite(top).all() and top.max() > 1.0


def test_loader_accepts_emissive_planes_and_spheres():
    body = """(scene :render_width 8 :render_height 8
      (camera :name 'c' :type 'standard')
      (emission :name 'l' :type 'standard' :radiance (illuminant 'd65'))
      (material :name 'm' :type 'diffuse')
      (entity :name 'p' :type 'plane' :material 'm' :emission 'l' :width 2 :height 1)
      (entity :name 's' :type 'sphere' :material 'm' :emission 'l' :radius 0.5))"""
    s = scene.PrcScene(source=body)
    d = s.desc
    assert d.n_entities == 2 and d.entities[0].kind == abi.ENTITY_PLANE and d.entities[0].emission == 0
    assert d.entities[1].kind == abi.ENTITY_SPHERE and d.entities[1].emission == 0
    o = ob.OracleScene(s)
    o.render(1)
    assert np.isfinite(o.output()[0]).all()
    with pytest.raises(RuntimeError, match="unknown emission"):
        scene.PrcScene(source=body.replace(":emission 'l' :radius", ":emission 'nope' :radius"))
