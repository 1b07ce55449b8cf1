"""The evaluation scene against an INDEPENDENT renderer: the reference tree ships examples/evaluation/cbox.exr, a Mitsuba 2 render of
examples/evaluation/scene.prc ("for the purpose of evaluating and comparing both implementations", its README).  tools/make_cbox_fixture.py
reduced it to 16 x 16 block means (tests/golden/cbox_mitsuba_16x16.json); our render of the same scene (pearray_amd.scene.cbox_eval, which
tests/test_prc_loader.py proves identical to loading scene.prc) must show the same picture.  This is a physical check of the whole pipeline
(geometry, camera, light transport, spectral upsampling, XYZ conversion), not a bit-level one: Mitsuba renders RGB, PearRay spectra, so
the bar is block luminance within a few percent."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from pearray_amd import scene
from pearray_amd import _cabi as abi

HERE = os.path.dirname(os.path.abspath(__file__))


def mitsuba_luminance():
    blocks = np.array(json.load(open(os.path.join(HERE, "golden", "cbox_mitsuba_16x16.json")))["blocks"])
    return blocks @ np.array([0.2126, 0.7152, 0.0722])  # Rec.709 luminance = CIE Y of linear sRGB


def compare(xyz, w):
    k = w // 16
    # single samples with a very large weight (unclamped path tracing) would otherwise decide the verdict of a whole block
    ya = np.minimum(xyz[..., 1], 2.0).reshape(16, k, 16, k).mean(axis=(1, 3))
    yb = mitsuba_luminance()
    off_light = yb < 1.0  # the blocks covering the emitter differ by sub-block alignment
    scale = (ya[off_light] * yb[off_light]).sum() / (yb[off_light] ** 2).sum()
    interior = np.linalg.norm((ya - yb)[2:14, 2:14][off_light[2:14, 2:14]]) / np.linalg.norm(yb[2:14, 2:14][off_light[2:14, 2:14]])
    everything = np.linalg.norm((ya - yb)[off_light]) / np.linalg.norm(yb[off_light])
    white = (ya / yb)[5:12, 5:8]  # blocks that lie entirely on white surfaces (back wall, floor, the tall box's front): no colour conversion involved
    return scale, interior, everything, white


def test_oracle_matches_the_mitsuba_render_of_the_evaluation_scene():
    w, spp = 64, 32
    o = ob.OracleScene(scene.cbox_eval(w, w, spp=spp))
    o.render(spp)
    scale, interior, everything, white = compare(o.output()[0].reshape(w, w, 3), w)
    assert 0.97 < scale < 1.10, scale          # overall energy (measured 1.03)
    assert abs(white.mean() - 1) < 0.05, white.mean()
    assert interior < 0.18 and everything < 0.20, (interior, everything)  # 0.12 / 0.15 at this sample count (Monte-Carlo noise included)


@pytest.mark.gpu
def test_gpu_render_matches_the_mitsuba_render_of_the_evaluation_scene():
    from pearray_amd import backend
    w, spp = 256, 128
    g = backend.RenderContext(scene.cbox_eval(w, w, spp=spp))
    g.start()
    g.waitForFinish()
    scale, interior, everything, white = compare(g.output()[0], w)
    assert 0.98 < scale < 1.09, scale
    # white surfaces agree with Mitsuba to a few percent block by block (1 % at 1024 spp); the coloured walls differ by 10-30 % in
    # luminance because Mitsuba renders RGB reflectances and PearRay the measured spectra
    assert np.abs(white - 1).max() < 0.08 and abs(white.mean() - 1) < 0.03, white
    assert interior < 0.10 and everything < 0.15, (interior, everything)  # measured 0.083 / 0.12 from 128 to 2048 spp: systematic, not noise


# ---- coloured spectra: a known answer of our own, and what an RGB view of the walls changes against Mitsuba ----------------------------
def _cie2006():
    """The renderer's own CIE 2006 2-degree tables (390..830 nm @ 1 nm), read from the product's table file as numbers."""
    import re
    txt = open(os.path.join(os.path.dirname(HERE), "pearray_amd", "csrc", "tables", "pr_tables.inl")).read()
    return [np.array([float(v.strip().rstrip("f")) for v in re.search(r"PR_CIE2006_%s\[441\] = \{(.*?)\};" % c, txt, re.S).group(1).split(",") if v.strip()]) for c in "XYZ"]


def _table(m, lam):
    v = np.asarray(m["values"], float)
    return np.interp((lam - m["start"]) / (m["end"] - m["start"]) * (len(v) - 1), np.arange(len(v)), v)   # constant beyond the ends, like EquidistantSpectrum


@pytest.mark.parametrize("material", ["white", "red", "green"])
def test_coloured_reflectance_under_a_coloured_light_has_the_colour_of_the_spectral_integral(material):
    """The form-factor scene of the reference's validity.py with MEASURED spectra: a diffuse plane with the evaluation scene's white / red /
    green reflectance under its luminaire's spectrum, direct light only.  Pixel XYZ = F(p) * integral(albedo * light * cmf) / integral(ybar):
    geometry (F, pinned by test_analytic_form_factor) times a one-line numeric integral -- a known answer for everything colour goes
    through (table lookup of both spectra, wavelength sampling and its pdf, the hero-wavelength weights, CIE accumulation), which the
    flat-spectrum tests cannot see and which the comparison with Mitsuba leaves open on the coloured walls."""
    from test_oracle_render import _form_factor
    import json
    data = json.load(open(os.path.join(os.path.dirname(HERE), "pearray_amd", "data", "cbox_eval.json")))
    W, spp = 24, 1024
    b = scene.SceneBuilder(W, W)
    s = b.settings
    s.aa_sampler, s.aa_samples, s.max_ray_depth, s.direct, s.filter, s.filter_radius, s.mapper = abi.SAMPLER_MJITT, spp, 1, 0, abi.FILTER_BLOCK, 0, abi.MAPPER_RANDOM
    m, e = data["materials"][material], data["emission"]
    plane = b.lambert(b.spectrum_table(m["start"], m["end"], m["values"]))
    ems = b.diffuse_emission(b.spectrum_table(e["start"], e["end"], e["values"]))
    b.add_mesh([[-2, -2, 0], [2, -2, 0], [2, 2, 0], [-2, 2, 0]], [[0, 1, 2], [0, 2, 3]], plane, normals=[[0, 0, 1]] * 4)
    b.add_mesh([[-0.5, -0.5, 2], [-0.5, 0.5, 2], [0.5, 0.5, 2], [0.5, -0.5, 2]], [[0, 1, 2], [0, 2, 3]], plane, emission=ems, normals=[[0, 0, -1]] * 4)
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = 1.0
    b.set_camera(T, width=2.0, height=2.0, local_direction=(0, 0, -1), local_right=(1, 0, 0), local_up=(0, 1, 0))
    o = ob.OracleScene(b.build())
    o.render(spp, threads=8)
    xyz, smp, fb = o.output()
    assert (fb == 0).all()
    lam = np.arange(390, 831, 1.0)
    X, Y, Z = _cie2006()
    product = _table(m, lam) * _table(e, lam)
    colour = np.array([(c * product).sum() for c in (X, Y, Z)]) / Y.sum()
    lo, hi = W // 2 - 4, W // 2 + 4
    F = np.mean([_form_factor((2 * (px / W - 0.5), -2 * (py / W - 0.5))) for py in range(lo, hi) for px in range(lo, hi)])
    got = xyz[lo:hi, lo:hi].reshape(-1, 3).mean(axis=0)
    assert np.allclose(got, F * colour, rtol=0.025, atol=0.004 * F * colour[1]), (material, got, F * colour)


def _cie1931():
    import re
    txt = open(os.path.join(os.path.dirname(HERE), "pearray_amd", "csrc", "tables", "pr_tables.inl")).read()
    return [np.array([float(v.strip().rstrip("f")) for v in re.search(r"PR_CIE1931_%s\[95\] = \{(.*?)\};" % c, txt, re.S).group(1).split(",") if v.strip()]) for c in "XYZ"]


def test_what_the_coloured_walls_difference_against_mitsuba_is_not():
    """The coloured walls differ from Mitsuba's image by a ratio that varies ALONG each wall (red: 0.88 at the front edge to 1.22 at the back;
    green: 0.92 at the back to 1.32 - 1.5 at the front; `tools/probe_cbox_walls.py`, `profiles/r04_cbox_walls.log`) while the white surfaces
    agree to 1 %.  Two explanations are excluded here, with numbers: (1) the colour matching functions -- first bounce analytically, sRGB of
    wall x light under CIE 1931 with the spectra zero outside 400..700 nm (Mitsuba) against CIE 2006 with constant extrapolation (PearRay):
    luminance within 3 % for white, red and green; (2) an RGB-mode render on Mitsuba's side -- three scalar renders with every spectrum
    replaced by its E-weighted sRGB channel, i.e. what an RGB renderer computes, show white surfaces 1.47 x too red against Mitsuba's image
    (its R : G there is 0.495, the spectral prediction 0.52, the RGB emulation 0.31): the image is a spectral render.  The path depth limit
    moves the wall ratios by < 0.05 (probe).  What remains is a difference between the two SCENES that the reference tree cannot settle
    (Mitsuba's scene file is not in it); our own coloured-spectra arithmetic is pinned by the known answer above."""
    import json
    data = json.load(open(os.path.join(os.path.dirname(HERE), "pearray_amd", "data", "cbox_eval.json")))
    M = np.array([[3.2404542, -1.5371385, -0.4985314], [-0.9692660, 1.8760108, 0.0415560], [0.0556434, -0.2040259, 1.0572252]])
    lum = lambda c: float(c @ np.array([0.2126, 0.7152, 0.0722]))

    def table(m, lam, zero_outside):
        v = _table(m, lam)
        return np.where((lam < m["start"]) | (lam > m["end"]), 0.0, v) if zero_outside else v

    X31, Y31, Z31 = _cie1931()
    X06, Y06, Z06 = _cie2006()
    l31, l06 = np.arange(360, 831, 5.0), np.arange(390, 831, 1.0)
    srgb31 = lambda s: M @ (np.array([(X31 * s).sum(), (Y31 * s).sum(), (Z31 * s).sum()]) / Y31.sum())
    for n in ("white", "red", "green"):
        a = srgb31(table(data["materials"][n], l31, True) * table(data["emission"], l31, True))
        s = table(data["materials"][n], l06, False) * table(data["emission"], l06, False)
        b = M @ (np.array([(X06 * s).sum(), (Y06 * s).sum(), (Z06 * s).sum()]) / Y06.sum())
        assert abs(lum(b) / lum(a) - 1) < 0.04, (n, lum(b) / lum(a))

    def scalar_render(c, w=64, spp=64):
        b = scene.SceneBuilder(w, w)
        s = b.settings
        s.aa_sampler, s.aa_samples, s.max_ray_depth, s.mapper, s.filter, s.filter_radius = abi.SAMPLER_SOBOL, spp, 6, abi.MAPPER_RANDOM, abi.FILTER_TRIANGLE, 0
        mats = {n: b.lambert(b.spectrum_const(float(np.clip(srgb31(table(m, l31, True))[c], 0, 1)))) for n, m in data["materials"].items()}
        ems = b.diffuse_emission(b.spectrum_const(float(max(srgb31(table(data["emission"], l31, True))[c], 0))))
        for ent in data["entities"]:
            T = np.eye(4, dtype=np.float32)
            if ent["position"]:
                T[:3, 3] = ent["position"]
            b.add_mesh(ent["p"], ent["faces"], mats[ent["material"]], normals=ent.get("n"), emission=ems if ent["emission"] else None, transform=T)
        cam = data["camera"]
        T = np.eye(4, dtype=np.float32)
        T[:3, 3] = cam["position"]
        b.set_camera(T, width=cam["width"][0], height=cam["height"][0], near=cam["near"][0], far=cam["far"][0], local_direction=cam["local_direction"],
                     local_right=cam["local_right"], local_up=cam["local_up"])
        o = ob.OracleScene(b.build())
        o.render(spp, threads=8)
        k = w // 16
        return np.minimum(o.output()[0].reshape(w, w, 3)[..., 1], 4.0).reshape(16, k, 16, k).mean(axis=(1, 3))   # constant spectra: Y is the scalar solution

    blocks = np.array(json.load(open(os.path.join(HERE, "golden", "cbox_mitsuba_16x16.json")))["blocks"])
    white = (slice(5, 12), slice(5, 8))
    red, green = scalar_render(0)[white], scalar_render(1)[white]
    assert (red / blocks[white][..., 0]).mean() > 1.3, (red / blocks[white][..., 0]).mean()            # measured 1.47: not an RGB-mode image
    assert abs((blocks[white][..., 1] / blocks[white][..., 0]).mean() - 0.495) < 0.02                  # Mitsuba's white surfaces, G : R
    assert (green / red).mean() < 0.36                                                                  # ... an RGB renderer's: 0.31
