"""Output side of the path: the EXR writer of the C ABI (host code, no GPU needed) and -- on the GPU -- the shading-point AOV
planes against the checker."""
import os
import struct

import numpy as np
import pytest

import oracle_binding as ob
from pearray_amd import backend, scene
from pearray_amd import _cabi as abi


def read_exr_uncompressed(path):
    """Independent reader of the subset prgpu_write_exr produces (OpenEXR file layout: scanline, NO_COMPRESSION, FLOAT)."""
    f = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", f, 0)
    assert magic == 20000630 and version == 2
    p, attrs = 8, {}
    while f[p] != 0:
        e = f.index(b"\0", p); name = f[p:e].decode(); p = e + 1
        e = f.index(b"\0", p); typ = f[p:e].decode(); p = e + 1
        (size,) = struct.unpack_from("<i", f, p); p += 4
        attrs[name] = (typ, f[p:p + size]); p += size
    p += 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    names, q, ch = [], 0, attrs["channels"][1]
    while ch[q] != 0:
        e = ch.index(b"\0", q); names.append(ch[q:e].decode())
        ptype, = struct.unpack_from("<i", ch, e + 1)
        assert ptype == 2
        q = e + 17
    assert names == sorted(names)
    offsets = struct.unpack_from("<%dQ" % h, f, p)
    out = {n: np.empty((h, w), np.float32) for n in names}
    for y in range(h):
        yy, nbytes = struct.unpack_from("<ii", f, offsets[y])
        assert yy == y and nbytes == 4 * w * len(names)
        row = np.frombuffer(f, np.float32, w * len(names), offsets[y] + 8).reshape(len(names), w)
        for k, n in enumerate(names):
            out[n][y] = row[k]
    return out


def test_exr_writer_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    chans = {"R": rng.random((17, 23), dtype=np.float32), "G": rng.random((17, 23), dtype=np.float32),
             "B": rng.random((17, 23), dtype=np.float32), "depth.Z": (rng.random((17, 23), dtype=np.float32) * 100)}
    chans["G"][3, 4] = np.inf
    path = str(tmp_path / "frame.exr")
    backend.write_exr(path, chans)
    back = read_exr_uncompressed(path)
    assert sorted(back) == ["B", "G", "R", "depth.Z"]
    for n in chans:
        assert np.array_equal(back[n], chans[n]), n
    assert abi.load().prgpu_write_exr(b"/nonexistent-dir/x.exr", 1, 1, 0, None, None, None) == -1


def test_exr_strides_select_components_of_an_interleaved_frame(tmp_path):
    import ctypes as C
    xyz = np.arange(5 * 4 * 3, dtype=np.float32).reshape(5, 4, 3)
    lib = abi.load()
    names = (C.c_char_p * 3)(b"X", b"Y", b"Z")
    base = xyz.ctypes.data
    planes = (C.POINTER(C.c_float) * 3)(*[C.cast(base + 4 * k, C.POINTER(C.c_float)) for k in range(3)])
    strides = (C.c_uint32 * 3)(3, 3, 3)
    path = str(tmp_path / "xyz.exr")
    assert lib.prgpu_write_exr(path.encode(), 4, 5, 3, names, planes, strides) == 0
    back = read_exr_uncompressed(path)
    for k, n in enumerate("XYZ"):
        assert np.array_equal(back[n], xyz[..., k])


@pytest.mark.gpu
def test_shading_point_aovs_match_the_checker_and_the_geometry():
    names = list(abi.AOV_NAMES)
    sc = scene.cornell_glassy(96, 96, spp=5, ior=1.5)
    g = backend.RenderContext(sc); g.enableAOVs(names); g.start(); g.waitForFinish()
    o = ob.OracleScene(sc); o.enable_aovs(names); o.render(5, threads=8)
    for n in names:
        assert np.array_equal(g.aov(n), o.aov(n)), n
    assert np.array_equal(g.output()[0], o.output()[0])            # enabling AOVs does not disturb the image
    smp = g.output()[1].astype(np.float32)
    hit = smp > 0
    nrm = g.aov("normal")[hit] / smp[hit][:, None]
    assert np.allclose(np.linalg.norm(g.aov("normal")[hit], axis=1) <= smp[hit] + 1e-3, True)      # sums of unit vectors
    depth = g.aov("depth")[hit] / smp[hit]
    pos = g.aov("position")[hit] / smp[hit][:, None]
    assert (depth > 0).all() and np.isfinite(pos).all() and np.isfinite(nrm).all()
    ent = g.aov("entity_id")[hit] / smp[hit]
    prim_ent = g.primaryHits()[0]
    same = np.isclose(ent, np.round(ent))                          # pixels whose samples all hit the same entity
    assert same.mean() > 0.8
    with pytest.raises(abi.PrgpuError):
        g.enableAOVs(["depth"])                                     # too late: iterations were rendered
    h = backend.RenderContext(sc)
    with pytest.raises(abi.PrgpuError):
        h.aov("depth")                                              # not enabled


def test_tone_mapper_modes():
    """ToneMapper::map (ToneMapper.cpp:12-79) with RGBConverter::fromXYZ (RGBConverter.cpp:15-24)."""
    xyz = np.array([[0.9505, 1.0, 1.089], [0.4124, 0.2126, 0.0193], [0.0, 0.0, 0.0], [0.2, -0.5, 0.1]], dtype=np.float32)
    rgb = backend.tonemap(xyz)
    assert np.allclose(rgb[0], [1, 1, 1], atol=2e-3) and np.allclose(rgb[1], [1, 0, 0], atol=2e-3)      # D65 white, the red primary
    assert (rgb >= 0).all() and rgb[3].min() == 0.0                                                      # clamped at zero
    m = np.array([[3.240970, -1.537383, -0.4986108], [-0.9692436, 1.875968, 0.04155506], [0.05563008, -0.2039770, 1.056972]], dtype=np.float32)
    assert np.allclose(rgb, np.maximum(0, xyz @ m.T), rtol=1e-6, atol=1e-7)
    assert np.array_equal(backend.tonemap(xyz, abi.TONE_XYZ), xyz)
    lum = backend.tonemap(xyz, abi.TONE_LUMINANCE)
    assert np.array_equal(lum, np.repeat(xyz[:, 1:2], 3, axis=1))
    w = np.array([2.0, 0.0, 1.0, 4.0], dtype=np.float32)
    half = backend.tonemap(xyz, abi.TONE_XYZ, scale=3.0, weight=w)
    assert np.allclose(half[0], xyz[0] / 2 * 3) and np.allclose(half[1], xyz[1] * 3)                      # weight <= eps leaves the pixel
    assert np.array_equal(backend.tonemap(xyz, abi.TONE_XYZ_NORM), np.zeros_like(xyz))                  # scales what the output holds (zeros here)
    lib = abi.load()
    assert lib.prgpu_tonemap(9, 1.0, backend._f32p(xyz), None, backend._f32p(rgb), 3, 4) == -1
    assert lib.prgpu_tonemap(0, 1.0, backend._f32p(xyz), None, backend._f32p(xyz), 3, 4) == -1            # in place


def test_output_blocks_are_parsed_like_the_reference():
    src = """(scene :render_width 8 :render_height 8
      (camera :name 'c' :type 'standard') (material :name 'm' :type 'diffuse')
      (mesh :name 'q' (attribute :type 'p' [0,0,0],[1,0,0],[0,1,0]) (faces [0,1,2]))
      (entity :name 'e' :type 'mesh' :mesh 'q' :materials 'm')
      (output :name 'image'
        (channel :type 'color' :color 'srgb') (channel :type 'n') (channel :type 'ng') (channel :type 'feedback')
        (channel :type 'color' :color 'srgb' :lpe 'CS*DL') (channel :type 'uv') (channel :type 'color' :lpe 'C((')
        (channel :type 'n' :lpe 'CDL') (channel :type 'color' :color 'xyz' :lpe 'C<TD"glass">*L'))
      (output :name 'extra' (channel :type 'RGB' :color 'XYZ') (channel :type 'var') (channel :type 'd') (channel :type 'samples') (channel :type 'nope')
        (channel :type 'feedback' :lpe 'CDL'))
      (output (channel :type 'color')))"""
    s = scene.PrcScene(source=src)
    ch, n = s.outputs()
    got = [(ch[i].file, ch[i].kind, ch[i].variable, ch[i].tone, ch[i].name.decode()) for i in range(n)]
    assert [ch[i].lpe.decode() for i in range(n)] == ["", "", "", "", "CS*DL", "", "CDL", 'C<TD"glass">*L'] + [""] * 4
    A = abi.AOV_NAMES.index
    assert got == [(0, abi.CHANNEL_SPECTRAL, 0, abi.TONE_SRGB, ""), (0, abi.CHANNEL_3D, A("normal"), 0, "normal"), (0, abi.CHANNEL_3D, A("normal_g"), 0, "normal_geometric"),
                   (0, abi.CHANNEL_COUNTER, 1, 0, "feedback"),
                   (0, abi.CHANNEL_SPECTRAL, 0, abi.TONE_SRGB, "[CS*DL]"),     # OutputSpecification.cpp:323-324
                   (0, abi.CHANNEL_SPECTRAL, 0, abi.TONE_SRGB, ""),            # an invalid expression is dropped, the channel stays (:299-302)
                   (0, abi.CHANNEL_3D, A("normal"), 0, "normal[CDL]"),        # a shading-point channel with an expression (:335-336)
                   (0, abi.CHANNEL_SPECTRAL, 0, abi.TONE_XYZ, '[C<TD"glass">*L]'), # a labelled token is legal (and matches nothing on this path)
                   (1, abi.CHANNEL_SPECTRAL, 0, abi.TONE_XYZ, ""), (1, abi.CHANNEL_SPECTRAL, 2, 0, "variance"), (1, abi.CHANNEL_1D, A("depth"), 0, "depth"),
                   (1, abi.CHANNEL_COUNTER, 0, 0, "sample_count")]
    lib = abi.load()
    assert lib.prgpu_prc_output_name(s._h, 0) == b"image" and lib.prgpu_prc_output_name(s._h, 1) == b"extra" and lib.prgpu_prc_output_name(s._h, 2) is None
    w = "\n".join(s.warnings)
    assert "invalid or unsupported light path expression 'C(('" in w and "colour and shading-point channels only" in w and "'uv' AOV" in w and "unknown channel type 'nope'" in w and "no name given" in w


@pytest.mark.gpu
@pytest.mark.parametrize("flt,r,mode", [(abi.FILTER_MITCHELL, 1, "persistent"), (abi.FILTER_MITCHELL, 1, "lockstep"), (abi.FILTER_MITCHELL, 1, "streaming"),
                                        (abi.FILTER_GAUSSIAN, 2, "persistent"), (abi.FILTER_GAUSSIAN, 2, "lockstep")])
def test_online_mean_and_variance_match_the_checker(monkeypatch, flt, r, mode):
    """Welford's update once per pixel and iteration (VarianceEstimator.inl:15-27) at the fold of every pipeline."""
    monkeypatch.setenv("PRGPU_MODE", mode)
    sc = scene.cornell_box(72, 60, spp=7, filter=flt, filter_radius=r)
    g = backend.RenderContext(sc); g.enableVariance(); g.render(3); g.render(4); g.waitForFinish()
    o = ob.OracleScene(sc); o.enable_variance(); o.render(7, threads=8)
    gm, gv = g.variance()
    om, ov = o.variance()
    if r == 1:   # single live tap: identical arithmetic
        assert np.array_equal(gm, om) and np.array_equal(gv, ov)
        assert np.array_equal(g.output()[0], o.output()[0])
    else:        # gather vs splat order of the taps
        assert np.allclose(gm, om, rtol=1e-4, atol=1e-7) and np.allclose(gv, ov, rtol=2e-3, atol=1e-7)
    assert gv.max() > 0 and (gv >= -1e-6).all()
    assert np.allclose(gm, g.output()[0], rtol=1e-4, atol=1e-6)        # the online mean IS the running mean of the frame
    with pytest.raises(abi.PrgpuError):
        g.enableVariance()


@pytest.mark.gpu
def test_output_blocks_end_to_end(tmp_path):
    """(output ...) of a .prc: planes enabled, rendered, written as EXR with the reference's channel names and weighting."""
    src = """(scene :render_width 40 :render_height 32 :camera 'c'
      (sampler :slot 'aa' :type 'sobol' :sample_count 4)
      (camera :name 'c' :type 'standard' :width 1 :height 0.8 :local_direction [0,0,-1] :local_up [0,1,0] :local_right [1,0,0] :position [0,1,4])
      (emission :name 'lamp' :type 'standard' :radiance (illum 6 6 5))
      (material :name 'white' :type 'diffuse' :albedo (refl 0.7 0.7 0.7))
      (mesh :name 'quad' (attribute :type 'p' [-1,0,-1],[1,0,-1],[1,0,1],[-1,0,1]) (faces [0,1,2,3]))
      (entity :name 'floor' :type 'mesh' :mesh 'quad' :materials 'white' :scale 2)
      (entity :name 'lamp' :type 'mesh' :mesh 'quad' :materials 'white' :emission 'lamp' :rotation (euler 180 0 0) :position [0,2,0] :scale 0.3)
      (output :name 'image' (channel :type 'color' :color 'srgb') (channel :type 'n') (channel :type 'feedback') (channel :type 'depth') (channel :type 'variance')
        (channel :type 'color' :color 'xyz' :lpe 'C.*L') (channel :type 'color' :color 'srgb' :lpe 'CDE') (channel :type 'color' :color 'xyz' :lpe 'CE')
        (channel :type 'ng' :lpe 'C.*') (channel :type 'p' :lpe 'CDL'))
      (output :name 'direct' (channel :type 'color' :color 'xyz' :lpe 'CDE')))"""
    s = scene.PrcScene(source=src)
    g = backend.RenderContext(s)
    g.enableOutputs(s)
    g.start(); g.waitForFinish()
    paths = g.saveOutputs(s, str(tmp_path))
    assert [os.path.basename(p) for p in paths] == ["image.exr", "direct.exr"]
    img = read_exr_uncompressed(paths[0])
    lpe_names = ["[%s].%s" % (e, c) for e in ("C.*L", "CDE", "CE") for c in "RGB"]
    aov_lpe = ["normal_geometric[C.*].%s" % c for c in "xyz"] + ["position[CDL].%s" % c for c in "xyz"]
    assert sorted(img) == sorted(["R", "G", "B", "variance.R", "variance.G", "variance.B", "normal.x", "normal.y", "normal.z", "depth", "feedback"] + lpe_names + aov_lpe)
    # shading points are pushed with the camera token only (direct.cpp:67,86-87): an expression that accepts "C" keeps the plane, any other empties it
    assert np.abs(img["normal_geometric[C.*].y"]).max() > 0.9 and not img["position[CDL].x"].any() and not img["position[CDL].z"].any()
    xyz, smp, fb = g.output()
    rgb = backend.tonemap(xyz)
    direct = read_exr_uncompressed(paths[1])
    assert sorted(direct) == sorted("[CDE]." + c for c in "RGB")              # the three distinct expressions are planes 0, 1, 2 in order of appearance
    for k, c in enumerate("RGB"):
        assert np.array_equal(img["[C.*L]." + c], xyz[..., k])               # every light path ('.' = any scattering, LPE_RegState.h:59-60): the frame itself
        assert np.array_equal(img["[CDE]." + c], backend.tonemap(g.lpe(1))[..., k]) and np.array_equal(direct["[CDE]." + c], g.lpe(1)[..., k])
        assert np.array_equal(img["[CE]." + c], g.lpe(2)[..., k])
    assert g.lpe(1)[..., 1].max() > 0 and g.lpe(2)[..., 1].sum() >= 0 and (g.lpe(1)[..., 1] <= xyz[..., 1] + 1e-6).all()
    for k, c in enumerate("RGB"):
        assert np.array_equal(img[c], rgb[..., k])
        assert np.array_equal(img["variance." + c], g.variance()[1][..., k])
    hit = smp > 0
    assert np.allclose(img["normal.y"][hit], (g.aov("normal")[..., 1] / np.maximum(smp, 1))[hit]) and np.allclose(np.abs(img["normal.y"][hit]), 1, atol=1e-5)
    assert np.allclose(img["depth"][hit], (g.aov("depth") / np.maximum(smp, 1))[hit]) and img["depth"][hit].min() > 1
    assert np.array_equal(img["feedback"], fb.astype(np.float32))
