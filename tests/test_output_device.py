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
