"""Host-side .prc loader (csrc/host/datalisp.cpp, prc_loader.cpp; C ABI prgpu_prc_*): syntax, semantic mapping onto
prgpu_scene_desc, error behaviour.  The checker is pearray_amd.scene.SceneBuilder (the Python scene assembly the parity tests
use) -- the loader must produce the same description, byte for byte, for the same scene."""
import ctypes as C
import os

import numpy as np
import pytest

from pearray_amd import _cabi as abi
from pearray_amd import scene

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = os.path.join(HERE, "golden", "scenes")
REF_EXAMPLES = "/root/reference/examples"


def arr(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).copy() if n else np.zeros(0, dtype)


def struct_bytes(x):
    return bytes(memoryview(x).cast("B")) if not isinstance(x, C.Structure) else C.string_at(C.addressof(x), C.sizeof(x))


def assert_same_desc(a, b, check_settings=True):
    """Both prgpu_scene_desc describe the same scene (pointers followed, padding-free structs compared bytewise)."""
    for f in ("n_vertices", "n_triangles", "n_entities", "n_materials", "n_emissions", "n_spectra", "n_spectral_table_values"):
        assert getattr(a, f) == getattr(b, f), f
    assert np.array_equal(arr(a.positions, 3 * a.n_vertices, np.float32), arr(b.positions, 3 * b.n_vertices, np.float32))
    assert bool(a.normals) == bool(b.normals)
    if a.normals:
        assert np.array_equal(arr(a.normals, 3 * a.n_vertices, np.float32), arr(b.normals, 3 * b.n_vertices, np.float32))
    assert bool(a.uvs) == bool(b.uvs)
    if a.uvs:
        assert np.array_equal(arr(a.uvs, 2 * a.n_vertices, np.float32), arr(b.uvs, 2 * b.n_vertices, np.float32))
    assert np.array_equal(arr(a.indices, 3 * a.n_triangles, np.uint32), arr(b.indices, 3 * b.n_triangles, np.uint32))
    assert np.array_equal(arr(a.tri_material, a.n_triangles, np.uint32), arr(b.tri_material, b.n_triangles, np.uint32))
    assert np.array_equal(arr(a.spectral_tables, a.n_spectral_table_values, np.float32), arr(b.spectral_tables, b.n_spectral_table_values, np.float32))
    assert a.n_lights == b.n_lights
    for name, n in (("entities", a.n_entities), ("materials", a.n_materials), ("emissions", a.n_emissions), ("spectra", a.n_spectra), ("lights", a.n_lights)):
        for i in range(n):
            assert struct_bytes(getattr(a, name)[i]) == struct_bytes(getattr(b, name)[i]), (name, i)
    assert struct_bytes(a.camera) == struct_bytes(b.camera)
    if check_settings:
        assert struct_bytes(a.settings) == struct_bytes(b.settings)


def two_quads_by_hand():
    """The scene of golden/scenes/two_quads.prc assembled through SceneBuilder."""
    b = scene.SceneBuilder(48, 32)
    s = b.settings
    s.spectral_start, s.spectral_end = 400.0, 700.0
    s.max_ray_depth, s.soft_max_ray_depth, s.mis = 8, 3, abi.MIS_POWER
    s.aa_sampler, s.aa_samples = abi.SAMPLER_SOBOL, 4
    s.filter, s.filter_radius = abi.FILTER_TRIANGLE, 1
    s.mapper = abi.MAPPER_SPD_HERO
    cam_t = np.eye(4, dtype=np.float32); cam_t[:3, 3] = [0, 0.5, 3]
    b.set_camera(cam_t, width=0.8, height=0.6, near=0.01, far=50, local_direction=(0, 0, -1), local_up=(0, 1, 0), local_right=(1, 0, 0))
    lamp = b.diffuse_emission(b.smul(b.illuminant_d65(), b.illum(4, 4, 3)))
    white = b.lambert(b.refl(0.7, 0.7, 0.7))
    red = b.lambert(b.refl(0.6, 0.1, 0.1), two_sided=False)
    grey = b.lambert(b.spectrum_const(0.5))
    tab = b.lambert(b.spectrum_table(400.0, 700.0, [0.1, 0.5, 0.9, 0.2]))
    fp = [[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1], [0, 0, 0]]
    b.add_mesh(fp, [[0, 3, 4], [3, 2, 4], [4, 2, 1], [0, 4, 1]], white, normals=[[0, 1, 0]] * 5, face_materials=[white, white, red, red],
               transform=np.diag([2, 1, 2, 1]).astype(np.float32), uvs=[[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5]])
    qp = [[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]]
    t = np.eye(4, dtype=np.float32)            # T * R(euler 180 about x) * S(0.5)
    c, sn = np.float32(np.cos(np.pi)), np.float32(np.sin(np.pi))
    return b, lamp, grey, tab, qp


def test_syntax_errors_are_reported_with_line_numbers():
    lib = abi.load()
    for src, needle in (("(scene :render_width 4", "missing ')'"), ("(scene ]", "mismatched"), ("scene", "expected '('"),
                        ("(scene :k 'abc)", "unterminated string"), ("(scene\n\n  (camera :name #))", "line 3"), ("( :a 1)", "identifier")):
        h = C.c_void_p()
        assert lib.prgpu_prc_load_string(src.encode(), None, None, C.byref(h)) == -1
        assert needle in lib.prgpu_prc_last_error().decode(), (src, lib.prgpu_prc_last_error())


def test_deep_nesting_is_a_parse_error_not_a_stack_overflow():
    lib = abi.load()
    h = C.c_void_p()
    src = "(scene " + "(a " * 100000
    assert lib.prgpu_prc_load_string(src.encode(), None, None, C.byref(h)) == -1
    assert "nesting too deep" in lib.prgpu_prc_last_error().decode()
    src = "(scene :k " + "[" * 100000
    assert lib.prgpu_prc_load_string(src.encode(), None, None, C.byref(h)) == -1
    assert "nesting too deep" in lib.prgpu_prc_last_error().decode()
    ok = "(scene :k " + "[" * 200 + "1" + "]" * 200 + ")"  # legal depth still parses (then fails semantically, not syntactically)
    lib.prgpu_prc_load_string(ok.encode(), None, None, C.byref(h))
    assert "nesting" not in lib.prgpu_prc_last_error().decode()


MINIMAL = """(scene :render_width 8 :render_height 8
  (camera :name 'c' :type 'standard')
  (material :name 'm' :type 'diffuse')
  (mesh :name 'q' (attribute :type 'p' [0,0,0],[1,0,0],[0,1,0]) (faces [0,1,2]))
  %s
  (entity :name 'e' :type 'mesh' :mesh 'q' :materials 'm')
)"""


def test_defaults_follow_the_reference():
    s = scene.PrcScene(source=MINIMAL % "")
    d, st = s.desc, s.desc.settings
    ref = abi.default_settings(8, 8)
    assert struct_bytes(st) == struct_bytes(ref)  # RenderSettings.cpp / manager defaults: sobol 128, mitchell r=1, spd cmis, depth 64/4
    cam = d.camera
    assert list(cam.local_direction) == [0, 1, 0] and list(cam.local_up) == [0, 0, 1] and list(cam.local_right) == [1, 0, 0]  # ICamera.cpp:5-7
    assert cam.width == 1 and cam.height == 1 and cam.near_t == np.float32(1e-6) and np.isinf(cam.far_t) and cam.fstop == 0
    assert d.materials[0].two_sided == 1 and d.spectra[d.materials[0].albedo].kind == abi.SPEC_CONST and d.spectra[d.materials[0].albedo].p[0] == 1.0
    assert d.entities[0].emission == abi.INVALID_ID and d.entities[0].has_normals == 0 and d.n_triangles == 1
    assert s.spp == 128


@pytest.mark.parametrize("block,code,needle", [
    ("(entity :name 's' :type 'subdiv' :mesh 'm')", -4, "entity type 'subdiv'"),
    ("(entity :name 's' :type 'sphere' :emission 'nope')", -1, "unknown emission"),
    ("(material :name 'g' :type 'glass' :roughness 'tex')", -4, "must be a number"),
    ("(material :name 'g' :type 'ward')", -4, "material type 'ward'"),
    ("(light :name 'sky' :type 'sky' :turbidity 0.5)", -1, "turbidities 1"),
    ("(light :name 'l' :type 'spot')", -4, "light type 'spot'"),
    ("(integrator :type 'vcm')", -4, "integrator 'vcm'"),
    ("(sampler :type 'blue_noise')", -4, "sampler type 'blue_noise'"),
    ("(filter :type 'box3')", -1, "unknown filter type 'box3'"),
    ("(spectral_mapper :type 'wide')", -4, "spectral mapper 'wide'"),
    ("(emission :name 'x' :type 'standard' :radiance (illuminant 'D93'))", -1, "unknown illuminant 'd93'"),
    ("(material :name 'x' :type 'diffuse' :albedo (perlin 1 2))", -4, "perlin"),
    ("(entity :name 'e2' :type 'mesh' :mesh 'nope' :materials 'm')", -1, "unknown mesh 'nope'"),
    ("(mesh :name 'bad' (attribute :type 'p' [0,0,0],[1,0,0],[0,1,0]) (faces [0,1,5]))", -1, "out of range"),
    ("(mesh :name 'bad' (attribute :type 'p' [0,0,0],[1,0,0],[0,1,0]) (faces [0,1,2,0,1]))", -1, "triangle or quad"),
    ("(include 'does_not_exist.inc')", -1, "cannot open"),
])
def test_unsupported_and_invalid_blocks_fail_loudly(block, code, needle):
    with pytest.raises(abi.PrgpuError) as e:
        scene.PrcScene(source=MINIMAL % block)
    assert e.value.args[1] == code and needle in e.value.args[0], e.value.args


def test_force_direct_and_overrides():
    s = scene.PrcScene(source=MINIMAL % "(integrator :type 'vcm' :max_ray_depth 6)", force_direct=True, width=20, height=10, spp=3, seed=7)
    st = s.desc.settings
    assert (st.width, st.height, st.aa_samples, st.seed, st.max_ray_depth) == (20, 10, 3, 7, 64)
    assert any("replaced by 'direct'" in w for w in s.warnings)


def test_two_quads_scene_matches_hand_assembly():
    s = scene.PrcScene(path=os.path.join(SCENES, "two_quads.prc"))
    d = s.desc
    ch, n = s.outputs()
    assert n == 1 and (ch[0].kind, ch[0].variable, ch[0].tone, ch[0].name) == (abi.CHANNEL_SPECTRAL, 0, abi.TONE_XYZ, b"")
    b, lamp, grey, tab, qp = two_quads_by_hand()
    # lamp: T(0,1.5,0) * Rx(180 deg) * S(0.5);   card: T * R(quaternion 45 deg about y) * S(0.3,1,0.3)
    got_lamp = np.array(list(d.entities[1].transform), dtype=np.float32).reshape(4, 4)
    want_lamp = np.array([[0.5, 0, 0, 0], [0, -0.5, 0, 1.5], [0, 0, -0.5, 0], [0, 0, 0, 1]], dtype=np.float32)
    assert np.allclose(got_lamp, want_lamp, atol=1e-6)
    got_card = np.array(list(d.entities[2].transform), dtype=np.float32).reshape(4, 4)
    r = np.sqrt(0.5)
    want_card = np.array([[r * 0.3, 0, r * 0.3, 0.5], [0, 1, 0, 0.3], [-r * 0.3, 0, r * 0.3, 0.2], [0, 0, 0, 1]], dtype=np.float32)
    assert np.allclose(got_card, want_card, atol=1e-6)
    b.add_mesh(qp, [[0, 1, 2, 3]], grey, emission=lamp, transform=got_lamp)   # rotation arithmetic checked above; reuse the exact matrices
    b.add_mesh(qp, [[0, 1, 2, 3]], tab, transform=got_card)
    want = b.build()  # keep the owner of the arrays alive while comparing
    assert_same_desc(d, want.desc)
    # quads split like Embree quads: (v0,v1,v3), (v2,v3,v1)
    idx = arr(d.indices, 3 * d.n_triangles, np.uint32).reshape(-1, 3)
    assert idx[4].tolist() == [5, 6, 8] and idx[5].tolist() == [7, 8, 6]
    assert arr(d.tri_material, d.n_triangles, np.uint32).tolist() == [0, 0, 1, 1, 2, 2, 3, 3]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_EXAMPLES, "cornellbox.prc")), reason="reference checkout not present (GPU box)")
def test_reference_cornellbox_prc_equals_the_committed_fixture_scene():
    """examples/cornellbox.prc (+ its include) parsed by the C++ loader == the scene assembled from pearray_amd/data/cornell_box.json
    (numbers extracted from the same file by tools/extract_fixtures.py).  The file asks for 'vcm': config C1 renders it with `direct`."""
    s = scene.PrcScene(path=os.path.join(REF_EXAMPLES, "cornellbox.prc"), force_direct=True, width=256, height=256, spp=16)
    want = scene.cornell_box(256, 256, spp=16)
    assert_same_desc(s.desc, want.desc, check_settings=False)
    st = s.desc.settings
    assert (st.width, st.height, st.aa_sampler, st.aa_samples) == (256, 256, abi.SAMPLER_MJITT, 16)
    assert struct_bytes(st) == struct_bytes(want.desc.settings)


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="reference checkout not present (GPU box)")
def test_entity_visibility_flags_are_accepted_and_reported_as_inert():
    """SceneLoader.cpp:452-497 stores the flags, Scene.cpp:135,166 traces every ray with MASK_ALL: they change nothing."""
    flagged = scene.PrcScene(source=MINIMAL % "(entity :name 's2' :type 'sphere' :radius 0.5 :material 'm' :camera_visible false :shadow_visible false)")
    plain = scene.PrcScene(source=MINIMAL % "(entity :name 's2' :type 'sphere' :radius 0.5 :material 'm')")
    assert sum("has no effect" in w for w in flagged.warnings) == 2 and not any("has no effect" in w for w in plain.warnings)
    assert flagged.desc.n_entities == plain.desc.n_entities and flagged.desc.n_triangles == plain.desc.n_triangles


def test_every_reference_example_either_loads_or_names_what_is_missing():
    lib = abi.load()
    loaded, refused = [], []
    table = np.full((256, 512, 11), 0.1, dtype=np.float32)
    sky = (abi.PrcSky * 1)(abi.PrcSky(None, table.ctypes.data_as(C.POINTER(C.c_float)), 512, 256))
    for name in sorted(os.listdir(REF_EXAMPLES)):
        if not name.endswith(".prc"):
            continue
        h = C.c_void_p()
        opt = abi.PrcOptions(0, 0, 0, 1, 0)
        opt.n_skies, opt.skies = 1, sky     # any sky light gets this (flat) table: the Hosek-Wilkie evaluation is the host's
        rc = lib.prgpu_prc_load_file(os.path.join(REF_EXAMPLES, name).encode(), C.byref(opt), C.byref(h))
        if rc == 0:
            loaded.append(name)
            lib.prgpu_prc_free(h)
        else:
            assert rc == -4, (name, rc, lib.prgpu_prc_last_error())   # valid DataLisp, unsupported feature -- never a syntax error
            msg = lib.prgpu_prc_last_error().decode()
            assert "not supported" in msg or "not available" in msg, name
            refused.append(name)
    assert "cornellbox.prc" in loaded and "area_lit_spheres.prc" in loaded and "complex.prc" in loaded and "material_array.prc" in loaded and "sky.prc" in loaded and "skylens.prc" in loaded and "box.prc" in loaded and "quadric_showcase.prc" in loaded and len(loaded) >= 18


def test_obj_embed_semantics(tmp_path):
    """v / vn / f with the four corner forms, negative indices, polygons fanned, per-corner expansion when normal indices differ."""
    (tmp_path / "m.obj").write_text("""# test mesh
o tetra
v 0 0 0
v 1 0 0
v 0 1 0
v 0 0 1
vn 0 0 -1
vn 0 -1 0
f 1//1 3//1 2//1
f 1//2 2//2 4//2
f -4/7/1 -1/8/1 -2/9/1
""")
    (tmp_path / "quad.obj").write_text("v -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nv 0 1 0\nf 1 2 3 4 5\n")
    src = MINIMAL % ("(embed :loader 'obj' :file 'm.obj') (embed :loader 'obj' :file 'quad.obj' :name 'pent')"
                     "(entity :name 't' :type 'mesh' :mesh 'tetra' :materials 'm') (entity :name 'p' :type 'mesh' :mesh 'pent' :materials 'm')")
    s = scene.PrcScene(source=src, include_dir=str(tmp_path))
    d = s.desc
    assert d.n_entities == 3 and d.entities[0].n_tris == 3 and d.entities[0].has_normals == 1 and d.entities[1].n_tris == 3 and d.entities[1].has_normals == 0
    pos = arr(d.positions, 3 * d.n_vertices, np.float32).reshape(-1, 3)
    nrm = arr(d.normals, 3 * d.n_vertices, np.float32).reshape(-1, 3)
    idx = arr(d.indices, 3 * d.n_triangles, np.uint32).reshape(-1, 3)
    # tetra: 9 expanded corners (normal index != vertex index); third face uses negative indices: v1, v4, v3 with normal 1
    assert idx[:3].tolist() == [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
    assert pos[6:9].tolist() == [[0, 0, 0], [0, 0, 1], [0, 1, 0]] and nrm[6:9].tolist() == [[0, 0, -1]] * 3 and nrm[3:6].tolist() == [[0, -1, 0]] * 3
    # pentagon: indexed, fanned around its first corner
    assert (idx[3:6] - 9).tolist() == [[0, 1, 2], [0, 2, 3], [0, 3, 4]]
    with pytest.raises(abi.PrgpuError) as e:
        scene.PrcScene(source=MINIMAL % "(embed :loader 'stl' :file 'm.stl')", include_dir=str(tmp_path))
    assert e.value.args[1] == -4


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_EXAMPLES, "evaluation", "scene.prc")), reason="reference checkout not present (GPU box)")
def test_reference_evaluation_scene_equals_the_committed_fixture_scene():
    """examples/evaluation/scene.prc (+ its eight OBJ meshes) through the C++ loader == pearray_amd/data/cbox_eval.json through SceneBuilder."""
    s = scene.PrcScene(path=os.path.join(REF_EXAMPLES, "evaluation", "scene.prc"))
    want = scene.cbox_eval()
    assert_same_desc(s.desc, want.desc)
    assert s.outputs()[1] >= 1


def test_plane_entity_matches_scene_builder():
    src = MINIMAL % ("(entity :name 'floor' :type 'plane' :material 'm' :x_axis [1,0,0] :y_axis [0,0,-1] :width 4 :height 3 :centering true :position [0,-1,0])"
                     "(entity :name 'wall' :type 'plane' :material 'm' :axis_x [0,2,0] :rotation (euler 0 90 0))")
    s = scene.PrcScene(source=src)
    b = scene.SceneBuilder(8, 8)
    b.set_camera(scene.IDENTITY, near=1e-6, local_direction=(0, 1, 0), local_up=(0, 0, 1), local_right=(1, 0, 0))
    m = b.lambert(b.spectrum_const(1.0))
    T = np.eye(4, dtype=np.float32); T[1, 3] = -1
    b.add_plane(m, x_axis=(1, 0, 0), y_axis=(0, 0, -1), width=4, height=3, centering=True, transform=T)
    b.add_plane(m, x_axis=(0, 2, 0), transform=np.array(list(s.desc.entities[1].transform), dtype=np.float32).reshape(4, 4))
    b.add_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]], m)
    want = b.build()
    assert_same_desc(s.desc, want.desc)
    d = s.desc
    assert d.entities[0].kind == abi.ENTITY_PLANE and d.entities[0].n_tris == 2 and d.entities[2].kind == abi.ENTITY_MESH
    pos = arr(d.positions, 12, np.float32).reshape(4, 3)
    assert pos.tolist() == [[-2, 0, 1.5], [-2, 0, -1.5], [2, 0, -1.5], [2, 0, 1.5]]          # p, p+y, p+y+x, p+x  (plane.cpp:81-84)
    assert arr(d.indices, 6, np.uint32).tolist() == [0, 1, 3, 2, 3, 1]
    lit = scene.PrcScene(source=MINIMAL % "(emission :name 'l' :type 'standard') (entity :name 'p' :type 'plane' :material 'm' :emission 'l')")
    assert lit.desc.entities[0].kind == abi.ENTITY_PLANE and lit.desc.entities[0].emission == 0   # emissive planes are area lights


def test_infinite_lights_match_scene_builder():
    src = MINIMAL % ("(light :name 'sky' :type 'env' :radiance (illuminant 'D65') :background 0.25 :rotation (euler 90 0 0))"
                     "(light :name 'sun' :type 'distant' :direction [0.2, -1, 0.1] :irradiance (illum 3 3 2))"
                     "(light :type 'background')")
    s = scene.PrcScene(source=src)
    b = scene.SceneBuilder(8, 8)
    b.set_camera(scene.IDENTITY, near=1e-6, local_direction=(0, 1, 0), local_up=(0, 0, 1), local_right=(1, 0, 0))
    m = b.lambert(b.spectrum_const(1.0))
    b.environment_light(b.illuminant_d65(), background=b.spectrum_const(0.25), transform=np.array(list(s.desc.lights[0].transform), dtype=np.float32).reshape(4, 4))
    b.distant_light(b.illum(3, 3, 2), direction=(0.2, -1, 0.1))
    b.environment_light(b.spectrum_const(1.0))
    b.add_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]], m)
    want = b.build()
    # the loader creates nodes in block order: material, then the lights; the builder above created the material first too
    assert_same_desc(s.desc, want.desc)
    rot = np.array(list(s.desc.lights[0].transform), dtype=np.float32).reshape(4, 4)
    assert np.allclose(rot[:3, :3], [[1, 0, 0], [0, 0, -1], [0, 1, 0]], atol=1e-6)


def test_sun_and_sky_lights_match_scene_builder():
    """(light :type 'sun') is tabulated by the loader (Preetham model, sun.cpp:42-46), (light :type 'sky') takes the host's table."""
    lib = abi.load()
    src = MINIMAL % ("(light :name 'sun' :type 'sun' :turbidity 2.5 :radius 4 :elevation 0.9 :azimuth 2.0 :power_scale 0.5)"
                     "(light :name 'sky' :type 'sky' :elevation_resolution 4 :azimuth_resolution 8 :extend false :rotation (euler 90 0 0))"
                     "(light :name 'dot' :type 'sun' :radius 0 :theta 0.4 :phi -1.0)")
    table = np.arange(4 * 8 * 11, dtype=np.float32).reshape(4, 8, 11) / 100
    own = scene.PrcScene(source=src)   # without a host table the loader builds the light's SkyModel itself (tests/test_hosek_sky.py)
    assert own.desc.lights[1].kind == abi.LIGHT_SKY and list(own.sky_params()) == [1]
    with pytest.raises(abi.PrgpuError, match="8 x 4"):
        scene.PrcScene(source=src, skies={"sky": np.zeros((5, 8, 11), np.float32)})
    s = scene.PrcScene(source=src, skies={"other": np.zeros((4, 8, 11), np.float32), "sky": table})
    b = scene.SceneBuilder(8, 8)
    b.set_camera(scene.IDENTITY, near=1e-6, local_direction=(0, 1, 0), local_up=(0, 0, 1), local_right=(1, 0, 0))
    m = b.lambert(b.spectrum_const(1.0))
    theta = np.float32(0.5) * np.float32(np.pi) - np.float32(0.9)
    sun = [lib.prgpu_sun_radiance(360.0 + i * ((760.0 - 360.0) / 63), float(theta), 2.5) for i in range(64)]
    b.sun_light((np.array(sun, np.float32) * np.float32(np.float32(0.5) / np.float32(16.0))).tolist(), 0.9, 2.0, radius=4.0)
    b.sky_light(table, extend=False, transform=np.array(list(s.desc.lights[1].transform), dtype=np.float32).reshape(4, 4))
    # radius 0: SunDeltaLight = a distant light whose irradiance is radiance * solid angle of the disc (sun.cpp:164-170)
    theta2 = np.float32(0.4)
    el2 = np.float32(0.5) * np.float32(np.pi) - theta2
    theta2b = np.float32(0.5) * np.float32(np.pi) - el2
    omega = np.float32(2) * np.float32(np.pi) * (np.float32(1) - np.cos(np.float32(np.float32(np.pi) / np.float32(180) * np.float32(0.5358) * np.float32(0.5))))
    dot = [np.float32(lib.prgpu_sun_radiance(360.0 + i * ((760.0 - 360.0) / 63), float(theta2b), 3.0)) * omega * np.float32(1.0) for i in range(64)]
    node = b.spectrum_table(360.0, 760.0, dot)
    az2 = np.float32(-1.0) + np.float32(2) * np.float32(np.pi)
    k = b.distant_light(node, direction=(np.sin(theta2b) * np.cos(az2), np.sin(theta2b) * np.sin(az2), np.cos(theta2b)))
    b.lights[k].flags = 4   # PRGPU_LIGHTF_SUN_DELTA
    b.add_mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]], m)
    want = b.build()
    d = s.desc
    assert d.lights[0].kind == abi.LIGHT_SUN and d.lights[1].kind == abi.LIGHT_SKY and d.lights[2].kind == abi.LIGHT_DISTANT
    assert d.lights[1].flags == 0 and (d.lights[1].azimuth_count, d.lights[1].elevation_count) == (8, 4)
    assert abs(d.lights[0].cos_theta - np.cos(np.deg2rad(0.5358) * 2)) < 1e-7
    for f in ("n_spectra", "n_spectral_table_values", "n_lights"):
        assert getattr(d, f) == getattr(want.desc, f), f
    got_t, want_t = arr(d.spectral_tables, d.n_spectral_table_values, np.float32), arr(want.desc.spectral_tables, d.n_spectral_table_values, np.float32)
    assert np.allclose(got_t, want_t, rtol=2e-6)   # the python side rounds products in a different order than the loader's floats
    assert np.array_equal(got_t[64:64 + table.size], table.reshape(-1))
    for i in range(3):
        for f in ("kind", "radiance", "background", "flags", "table_offset", "azimuth_count", "elevation_count"):
            assert getattr(d.lights[i], f) == getattr(want.desc.lights[i], f), (i, f)
        assert np.allclose(list(d.lights[i].direction), list(want.desc.lights[i].direction), atol=1e-6)
        assert abs(d.lights[i].cos_theta - want.desc.lights[i].cos_theta) < 1e-7
    # the sun's spectrum is physically plausible: tens of thousands of W / (m^2 nm sr) at noon, redder towards the horizon
    hi = np.array([lib.prgpu_sun_radiance(w, 0.2, 3.0) for w in (450.0, 650.0)])
    lo = np.array([lib.prgpu_sun_radiance(w, 1.45, 3.0) for w in (450.0, 650.0)])
    assert 1.0e4 < hi[0] < 3.5e4 and lo[0] / lo[1] < 0.5 * hi[0] / hi[1]


def test_sun_position_defaults_to_the_reference_example_value():
    """SunLocation.h:7-8: 'Default is Saarbruecken 2020.05.06 12:00:00 (midday) which results in Elevation: 52.87 Azimuth: 143.27'."""
    lib = abi.load()
    el, az = C.c_float(), C.c_float()
    lib.prgpu_sun_position(2020, 5, 6, 12, 0, 0.0, 49.235422, 6.9965744, 2.0, C.byref(el), C.byref(az))
    assert abs(np.rad2deg(el.value) - 52.87) < 0.05 and abs(np.rad2deg(az.value) - 143.27) < 0.05   # the comment rounds


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="reference checkout absent")
def test_complex_prc_loads_with_a_host_supplied_sky_table():
    """BASELINE config C5: examples/complex.prc needs the SkyModel table of its 'sky' light (512 x 256 cells x 11 bands)."""
    table = np.full((256, 512, 11), 0.5, dtype=np.float32)
    s = scene.PrcScene(path=os.path.join(REF_EXAMPLES, "complex.prc"), skies={"sky": table})
    d = s.desc
    assert (d.settings.width, d.settings.height, d.settings.aa_samples, d.settings.filter_radius) == (1920, 1080, 4096, 0)
    kinds = sorted(d.lights[i].kind for i in range(d.n_lights))
    assert kinds == [abi.LIGHT_SKY, abi.LIGHT_SUN]
    sun = [d.lights[i] for i in range(d.n_lights) if d.lights[i].kind == abi.LIGHT_SUN][0]
    assert abs(sun.cos_theta - np.cos(np.deg2rad(0.5358) * 2)) < 1e-7        # :radius 4
    el, az = C.c_float(), C.c_float()
    abi.load().prgpu_sun_position(2020, 5, 6, 16, 0, 0.0, 49.235422, 6.9965744, 2.0, C.byref(el), C.byref(az))   # :hour 16
    assert np.allclose(list(sun.direction), [np.cos(el.value) * np.cos(az.value), np.cos(el.value) * np.sin(az.value), np.sin(el.value)], atol=1e-6)
    mats = sorted(d.materials[i].kind for i in range(d.n_materials))
    assert abi.MAT_PRINCIPLED in mats and abi.MAT_DIELECTRIC in mats and abi.MAT_LAMBERT in mats
    assert sum(d.entities[i].kind == abi.ENTITY_SPHERE for i in range(d.n_entities)) == 4 and d.n_triangles > 50000
    ch, n = s.outputs()   # complex.prc:27-34: color (srgb), n, ng, feedback; the commented-out LPE channel is not there
    assert [(ch[i].kind, ch[i].name) for i in range(n)] == [(abi.CHANNEL_SPECTRAL, b""), (abi.CHANNEL_3D, b"normal"), (abi.CHANNEL_3D, b"normal_geometric"),
                                                           (abi.CHANNEL_COUNTER, b"feedback")]


def test_scene_cache_round_trips_a_description_and_drops_sky_tables(tmp_path):
    table = (np.arange(4 * 8 * 11, dtype=np.float32).reshape(4, 8, 11) + 1) / 50
    src = MINIMAL % ("(light :name 'sky' :type 'sky' :elevation_resolution 4 :azimuth_resolution 8)"
                     "(light :name 'sun' :type 'sun' :radius 2 :elevation 0.7 :azimuth 1.0)"
                     "(material :name 'g' :type 'glass' :index (lookup_index 'bk7'))")
    s = scene.PrcScene(source=src, skies={"sky": table})
    path = str(tmp_path / "cache.npz")
    scene.save_scene_npz(path, s.desc)
    assert os.path.getsize(path) < 20000
    back = scene.ArrayScene(path, sky_tables=[table])
    a, b = s.desc, back.desc
    # tables are reordered (the sky's go last), so compare what the nodes see
    for f in ("n_vertices", "n_triangles", "n_entities", "n_materials", "n_spectra", "n_lights", "n_spectral_table_values"):
        assert getattr(a, f) == getattr(b, f), f
    ta, tb = arr(a.spectral_tables, a.n_spectral_table_values, np.float32), arr(b.spectral_tables, b.n_spectral_table_values, np.float32)
    for i in range(a.n_spectra):
        x, y = a.spectra[i], b.spectra[i]
        assert (x.kind, x.table_count) == (y.kind, y.table_count)
        assert np.array_equal(ta[x.table_offset:x.table_offset + x.table_count], tb[y.table_offset:y.table_offset + y.table_count])
    la, lb = a.lights[0], b.lights[0]
    n = 4 * 8 * 11
    assert np.array_equal(ta[la.table_offset:la.table_offset + n], tb[lb.table_offset:lb.table_offset + n])
    assert struct_bytes(a.camera) == struct_bytes(b.camera) and struct_bytes(a.settings) == struct_bytes(b.settings)
    assert np.array_equal(arr(a.positions, 9, np.float32), arr(b.positions, 9, np.float32))
    with pytest.raises(AssertionError, match="sky table shape"):
        scene.ArrayScene(path, sky_tables=[np.zeros((5, 8, 11), np.float32)])


def test_c5_fixture_is_the_reference_scene():
    """tests/golden/scenes/complex_c5.npz (tools/make_c5_fixture.py) against a fresh load of examples/complex.prc."""
    path = os.path.join(HERE, "golden", "scenes", "complex_c5.npz")
    fx = scene.ArrayScene(path)   # the sky table is rebuilt from the stored sun position, turbidity and ground albedo
    assert (fx.desc.n_triangles, fx.desc.n_entities, fx.desc.n_materials, fx.desc.n_lights) == (304046, 68, 6, 2)
    sky = [fx.desc.lights[i] for i in range(fx.desc.n_lights) if fx.desc.lights[i].kind == abi.LIGHT_SKY][0]
    table = fx.tables[sky.table_offset:sky.table_offset + 256 * 512 * 11]
    assert np.isfinite(table).all() and table.min() >= 0 and 0.1 < table.max() < 1.0   # W / (m^2 sr nm): the circumsolar peak of a clear sky
    if not os.path.isdir(REF_EXAMPLES):
        return
    ref = scene.PrcScene(path=os.path.join(REF_EXAMPLES, "complex.prc"))   # loads as shipped: no host-supplied table
    rsky = [ref.desc.lights[i] for i in range(ref.desc.n_lights) if ref.desc.lights[i].kind == abi.LIGHT_SKY][0]
    assert np.array_equal(arr(ref.desc.spectral_tables, ref.desc.n_spectral_table_values, np.float32)[rsky.table_offset:rsky.table_offset + 256 * 512 * 11], table)
    for f in ("n_vertices", "n_triangles", "n_entities", "n_materials", "n_spectra", "n_lights"):
        assert getattr(fx.desc, f) == getattr(ref.desc, f), f
    assert np.array_equal(arr(fx.desc.positions, 3 * fx.desc.n_vertices, np.float32), arr(ref.desc.positions, 3 * ref.desc.n_vertices, np.float32))
    assert np.array_equal(arr(fx.desc.indices, 3 * fx.desc.n_triangles, np.uint32), arr(ref.desc.indices, 3 * ref.desc.n_triangles, np.uint32))
    for i in range(fx.desc.n_materials):
        assert struct_bytes(fx.desc.materials[i]) == struct_bytes(ref.desc.materials[i])


EMBED = """(scene :render_width 8 :render_height 8
  (camera :name 'c' :type 'standard')
  (material :name 'm' :type 'diffuse')
  (embed :loader '%s' :file '%s' :name 'shape' %s)
  (entity :name 'e' :type 'mesh' :mesh 'shape' :materials 'm'))"""


def _embed_reference(positions, faces, normals=None, uvs=None):
    b = scene.SceneBuilder(8, 8)
    b.set_camera(scene.IDENTITY, near=1e-6, local_direction=(0, 1, 0), local_up=(0, 0, 1), local_right=(1, 0, 0))
    m = b.lambert(b.spectrum_const(1.0))
    b.add_mesh(positions, faces, m, normals=normals, uvs=uvs)
    return b.build()


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
def test_ply_embed_semantics(tmp_path, fmt):
    """PlyLoader.cpp: float properties by position (an unknown one is skipped), normals normalised on load, triangles and quads."""
    import struct
    P = [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.5, 0.5, 1]]
    N = [[0, 0, 2], [0, 0, 1], [0, 3, 0], [0, 0, 0], [1, 1, 1]]           # unnormalised; a zero normal stays zero (norm := 1)
    UV = [[0, 0], [1, 0], [1, 1], [0, 1], [0.5, 0.5]]
    F = [[0, 1, 2, 3], [0, 1, 4], [1, 2, 4]]
    header = "ply\nformat %s 1.0\ncomment made by a test\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\nproperty float confidence\n" \
             "property float nx\nproperty float ny\nproperty float nz\nproperty float u\nproperty float v\nelement face 3\nproperty list uchar int vertex_indices\nend_header\n" % fmt
    path = tmp_path / "m.ply"
    if fmt == "ascii":
        body = "".join("%g %g %g 0.5 %g %g %g %g %g\n" % (*p, *n, *uv) for p, n, uv in zip(P, N, UV)) + "".join("%d %s\n" % (len(f), " ".join(map(str, f))) for f in F)
        path.write_text(header + body)
    else:
        e = "<" if "little" in fmt else ">"
        body = b"".join(struct.pack(e + "9f", *p, 0.5, *n, *uv) for p, n, uv in zip(P, N, UV)) + b"".join(struct.pack(e + "B%di" % len(f), len(f), *f) for f in F)
        path.write_bytes(header.encode() + body)
    s = scene.PrcScene(source=EMBED % ("ply", str(path), ""))
    Nn = [np.array(n, np.float32) / (np.float32(np.sqrt(np.float32(sum(np.float32(c) * np.float32(c) for c in n)))) or np.float32(1)) for n in N]
    want = _embed_reference(P, F, normals=Nn, uvs=UV)
    assert_same_desc(s.desc, want.desc)
    assert s.desc.n_triangles == 4 and s.desc.entities[0].has_uvs == 1
    with pytest.raises(abi.PrgpuError, match="not a ply file"):
        (tmp_path / "bad.ply").write_text("plx\n")
        scene.PrcScene(source=EMBED % ("ply", str(tmp_path / "bad.ply"), ""))
    with pytest.raises(abi.PrgpuError, match="embed loader 'stl'"):
        scene.PrcScene(source=EMBED % ("stl", str(path), ""))


@pytest.mark.parametrize("version,double,with_normals", [(4, False, True), (3, True, False), (4, False, False)])
def test_mitsuba_serialized_embed_semantics(tmp_path, version, double, with_normals):
    """MtsSerializedLoader.cpp: zlib streams, end-of-file shape dictionary (u64 offsets, u32 in version 3), :shape selects one;
    meshes without normals get MeshBase::buildSmoothNormals (the last face of a vertex wins, MeshBase.cpp:15-60)."""
    import struct, zlib
    shapes = [([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]]),
              ([[0, 0, 0], [2, 0, 0], [2, 2, 0], [0, 2, 0], [1, 1, 3]], [[0, 1, 2], [0, 2, 3], [0, 1, 4], [2, 3, 4]])]
    # shape k's record starts with its own 4-byte header; offsets are absolute file positions of those headers
    recs, blob = [], b""
    for k, (P, F) in enumerate(shapes):
        recs.append(len(blob))
        flags = (0x2000 if double else 0x1000) | (0x0001 if with_normals else 0) | 0x0008
        raw = struct.pack("<I", flags) + (b"shape%d\0" % k if version >= 4 else b"") + struct.pack("<QQ", len(P), len(F))
        ft = "d" if double else "f"
        raw += struct.pack("<%d%s" % (3 * len(P), ft), *[c for p in P for c in p])
        if with_normals:
            raw += struct.pack("<%d%s" % (3 * len(P), ft), *([0.0, 0.0, 1.0] * len(P)))
        raw += struct.pack("<%d%s" % (3 * len(P), ft), *([0.5] * 3 * len(P)))
        raw += struct.pack("<%dI" % (3 * len(F)), *[i for f in F for i in f])
        blob += struct.pack("<HH", 0x041C, version) + zlib.compress(raw)
    blob += b"".join(struct.pack("<Q" if version >= 4 else "<I", r) for r in recs) + struct.pack("<I", len(shapes))
    path = tmp_path / "m.serialized"
    path.write_bytes(blob)
    for k, (P, F) in enumerate(shapes):
        s = scene.PrcScene(source=EMBED % ("mts", str(path), ":shape %d" % k))
        if with_normals:
            N = [[0, 0, 1]] * len(P)
        else:
            N = np.zeros((len(P), 3), np.float32)
            Pa = np.array(P, np.float32)
            for f in F:
                n = np.cross(Pa[f[1]] - Pa[f[0]], Pa[f[2]] - Pa[f[0]]).astype(np.float32)
                n = n / np.float32(np.sqrt(np.float32(n @ n)))
                for i in f:
                    N[i] = n
        want = _embed_reference(P, F, normals=N)
        d, w = s.desc, want.desc
        assert (d.n_vertices, d.n_triangles) == (w.n_vertices, w.n_triangles)
        assert np.array_equal(arr(d.positions, 3 * d.n_vertices, np.float32), arr(w.positions, 3 * w.n_vertices, np.float32))
        assert np.array_equal(arr(d.indices, 3 * d.n_triangles, np.uint32), arr(w.indices, 3 * w.n_triangles, np.uint32))
        assert np.allclose(arr(d.normals, 3 * d.n_vertices, np.float32), arr(w.normals, 3 * w.n_vertices, np.float32), atol=1e-6)
    with pytest.raises(abi.PrgpuError, match="cannot access shape 2"):
        scene.PrcScene(source=EMBED % ("mts", str(path), ":shape 2"))
    (tmp_path / "bad.serialized").write_bytes(b"\x00\x00\x04\x00" + b"\0" * 16)
    with pytest.raises(abi.PrgpuError, match="not a valid Mitsuba"):
        scene.PrcScene(source=EMBED % ("mts", str(tmp_path / "bad.serialized"), ""))
    # counts taken from the file are checked against the stream before anything is sized by them (a hostile file must not ask for 48 GB)
    raw = struct.pack("<I", 0x1000) + (b"x\0" if version >= 4 else b"") + struct.pack("<QQ", 0xFFFFFFFF, 1) + b"\0" * 64
    huge = struct.pack("<HH", 0x041C, version) + zlib.compress(raw) + struct.pack("<Q" if version >= 4 else "<I", 0) + struct.pack("<I", 1)
    (tmp_path / "huge.serialized").write_bytes(huge)
    with pytest.raises(abi.PrgpuError, match="announces 4294967295 vertices"):
        scene.PrcScene(source=EMBED % ("mts", str(tmp_path / "huge.serialized"), ""))
    (tmp_path / "count.serialized").write_bytes(blob[:-4] + struct.pack("<I", 0x7FFFFFFF))   # a shape count the file cannot hold
    with pytest.raises(abi.PrgpuError, match="does not fit the file"):
        scene.PrcScene(source=EMBED % ("mts", str(tmp_path / "count.serialized"), ""))


def _spd_scene(tmp_path, csv_text, expr="(spd 'data.csv')"):
    (tmp_path / "data.csv").write_text(csv_text)
    src = "(scene (camera :name 'c' :type 'standard') (emission :name 'e' :type 'standard' :radiance %s) (material :name 'm' :type 'diffuse'))" % expr
    s = scene.PrcScene(source=src, include_dir=str(tmp_path))
    sp = s.desc.spectra[s.desc.emissions[0].radiance]
    values = [s.desc.spectral_tables[sp.table_offset + i] for i in range(sp.table_count)]
    return s, sp, values


def test_spd_file_node_reads_csv_like_the_reference(tmp_path):
    """SPDFilePlugin::create (node/SPDFileNode.cpp:20-91) over CSV::read (base/container/CSV.cpp:52-163; its behaviours as exercised by
    src/tests/csv.cpp: header detection, ',' and ';', empty tokens skipped, invalid numbers -> 0, short lines dropped)."""
    s, sp, v = _spd_scene(tmp_path, "Wavelength;Rel.Power\n400;1.5\n410;2.5\n420;-3\n430;4e2\n")
    assert (sp.kind, sp.table_count, sp.wl_start, sp.wl_end) == (abi.SPEC_TABLE, 4, 400.0, 430.0) and v == [1.5, 2.5, 0.0, 400.0]   # clamped at zero (:82)
    _, sp, v = _spd_scene(tmp_path, "400, 1, 10\n500, 2, 20,\n600,, 3, 30\n700, x, 40\n", "(spd 'data.csv' 2 true)")       # no header; column 2 in percent
    assert (sp.table_count, sp.wl_start, sp.wl_end) == (4, 400.0, 700.0) and np.allclose(v, [0.1, 0.2, 0.3, 0.4])
    _, sp, v = _spd_scene(tmp_path, "400, 1, 10\n500, 2, 20,\n600,, 3, 30\n700, x, 40\n")                                    # "x" is not a number: 0
    assert v == [1.0, 2.0, 3.0, 0.0]
    _, sp, v = _spd_scene(tmp_path, "l,p\n400,1\n450\n500,2\n600,3,9\n", "(spd :file 'data.csv' :column 1 :percentage false)")  # the short line is dropped; extra tokens ignored
    assert (sp.table_count, sp.wl_start, sp.wl_end) == (3, 400.0, 600.0) and v == [1.0, 2.0, 3.0]
    s, sp, v = _spd_scene(tmp_path, "400,1\n500,2\n650,3\n")
    assert any("non equidistant" in w for w in s.warnings) and (sp.wl_start, sp.wl_end) == (400.0, 650.0)
    for text, expr, needle in (("400,1\n", "(spd 'data.csv')", "not enough data"), ("400\n500\n", "(spd 'data.csv')", "not enough data"),
                               ("500,1\n400,2\n", "(spd 'data.csv')", "invalid wavelengths"), ("400,1\n500,2\n", "(spd 'data.csv' 2)", "column 2"),
                               ("400,1\n500,2\n", "(spd 'missing.csv')", "not valid"), ("400,1\n500,2\n", "(spd)", "no file")):
        with pytest.raises(abi.PrgpuError, match=needle):
            _spd_scene(tmp_path, text, expr)


def test_reference_uv_light_spectrum_loads():
    path = os.path.join(REF_EXAMPLES, "UVLight.csv") if "REF_EXAMPLES" in globals() else "/root/reference/examples/UVLight.csv"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    src = "(scene (camera :name 'c' :type 'standard') (emission :name 'e' :type 'standard' :radiance (smul (spd 'UVLight.csv') 100)) (material :name 'm' :type 'diffuse'))"
    s = scene.PrcScene(source=src, include_dir=os.path.dirname(path))            # cornellbox_fluorescent.prc:56
    mul = s.desc.spectra[s.desc.emissions[0].radiance]
    assert mul.kind == abi.SPEC_MUL
    sp = s.desc.spectra[mul.lhs] if s.desc.spectra[mul.lhs].kind == abi.SPEC_TABLE else s.desc.spectra[mul.rhs]
    v = np.array([s.desc.spectral_tables[sp.table_offset + i] for i in range(sp.table_count)])
    assert (sp.table_count, sp.wl_start, sp.wl_end) == (111, 300.0, 410.0) and v.max() > 0 and (v >= 0).all() and 340 < 300 + v.argmax() < 380


def test_every_reference_illuminant_is_available():
    """IlluminantNode.cpp:89-126: d65, d50, d55, d75, a, c over 300..830 nm (107 samples), f1..f12 over 380..780 nm (81 samples), e = 1.
    The tables are the reference's arrays as declared (pearray_amd/csrc/tables/pr_illuminants.inl, generated by tools/extract_fixtures.py)."""
    for name, (count, start, end) in [(n, (107, 300.0, 830.0)) for n in ("D65", "d50", "D55", "d75", "A", "c")] + [("f%d" % k, (81, 380.0, 780.0)) for k in range(1, 13)]:
        src = "(scene (camera :name 'c' :type 'standard') (emission :name 'e' :type 'standard' :radiance (illuminant '%s')) (material :name 'm' :type 'diffuse'))" % name
        s = scene.PrcScene(source=src)
        sp = s.desc.spectra[s.desc.emissions[0].radiance]
        v = np.array([s.desc.spectral_tables[sp.table_offset + i] for i in range(sp.table_count)])
        assert (sp.kind, sp.table_count, sp.wl_start, sp.wl_end) == (abi.SPEC_TABLE, count, start, end), name
        assert (v >= 0).all() and v.max() > 0, name
        if name.lower() in ("d50", "d55", "d75", "c"):                 # daylight-like: unit scale around 560 nm
            assert 0.8 < v[(560 - 300) // 5] < 1.2, (name, v[52])
    s = scene.PrcScene(source="(scene (camera :name 'c' :type 'standard') (emission :name 'e' :type 'standard' :radiance (illuminant :spectrum 'E')) (material :name 'm' :type 'diffuse'))")
    assert s.desc.spectra[s.desc.emissions[0].radiance].kind == abi.SPEC_CONST
    with pytest.raises(abi.PrgpuError, match="unknown illuminant"):
        scene.PrcScene(source="(scene (camera :name 'c' :type 'standard') (emission :name 'e' :type 'standard' :radiance (illuminant 'd93')) (material :name 'm' :type 'diffuse'))")
