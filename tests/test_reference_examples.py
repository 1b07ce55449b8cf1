"""Every reference example the loader accepts (17 of examples/*.prc next to complex.prc = BASELINE config C5, which has its own fixture and tests; `vcm` / `ao` integrators replaced by `direct`), reduced to array
fixtures by tools/make_example_fixtures.py (the reference's files do not travel to the GPU box): the HIP path renders each one and is
compared with the checker -- hit ids, sample / feedback planes and statistics exact, XYZ bit for bit where the pixel filter has a single
live tap.  CPU: the fixtures are current (equal to a fresh conversion where the reference checkout exists) and the checker renders them."""
import glob
import os

import numpy as np
import pytest

from pearray_amd import _cabi as abi
from pearray_amd import scene
import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "scenes", "examples", "*.npz")))
NAMES = [os.path.basename(f)[:-4] for f in FIXTURES]
REF_EXAMPLES = "/root/reference/examples"


def load(path):
    return scene.ArrayScene(path)   # sky lights: the Hosek-Wilkie tables are rebuilt from the stored parameters (prgpu_sky_table)


def single_tap(sc):
    s = sc.desc.settings
    return s.filter_radius == 0 or s.filter == abi.FILTER_BLOCK or (s.filter == abi.FILTER_MITCHELL and s.filter_radius == 1)


def test_there_is_a_fixture_for_every_example_the_loader_accepts():
    assert len(FIXTURES) == 17 and "cornellbox" in NAMES and "material_showcase" in NAMES and "skylens" in NAMES
    if not os.path.isdir(REF_EXAMPLES):
        pytest.skip("reference checkout not present")
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import make_example_fixtures as mk
    accepted = []
    for f in sorted(glob.glob(os.path.join(REF_EXAMPLES, "*.prc"))):
        name = os.path.basename(f)
        if name in mk.SKIP:
            continue
        try:
            s = mk.load(f)
        except abi.PrgpuError:
            continue
        accepted.append(name[:-4])
        import tempfile
        with tempfile.TemporaryDirectory() as d:          # the committed fixture is what the loader produces today
            p = os.path.join(d, "x.npz")
            scene.save_scene_npz(p, s.desc, sky_params=s.sky_params())
            a, b = np.load(p), np.load(os.path.join(HERE, "golden", "scenes", "examples", name[:-4] + ".npz"))
            assert sorted(a.files) == sorted(b.files), name
            for k in a.files:
                assert np.array_equal(a[k], b[k]), (name, k)
    assert accepted == NAMES


@pytest.mark.parametrize("name", NAMES)
def test_checker_renders_the_example(name):
    sc = load(FIXTURES[NAMES.index(name)])
    sc.desc.settings.width, sc.desc.settings.height = 32, 24
    o = ob.OracleScene(sc)
    o.render(2, threads=8)
    xyz, smp, fb = o.output()
    assert np.isfinite(xyz).all() and (xyz >= 0).all()
    st = o.statistics()
    assert st["pixel_samples"] == 32 * 24 * 2 and st["camera_rays"] > 0
    if name not in ("skylens",):                           # an entity-less scene has nothing to hit
        assert st["entity_hits"] > 0 and smp.sum() > 0, st   # the sample plane counts shading points (commitShadingPoints)
    if name not in ("mesh",):                              # an ambient-occlusion scene without any emitter: black under `direct`
        assert xyz.sum() > 0, "a black frame"


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_renders_the_example_like_the_checker(name):
    from pearray_amd import backend
    sc = load(FIXTURES[NAMES.index(name)])
    iters = 4
    g = backend.RenderContext(sc)
    g.render(iters); g.waitForFinish()
    o = ob.OracleScene(sc)
    o.render(iters, threads=16)
    gx, gs, gf = g.output()
    ox, os_, of = o.output()
    ge, gp = g.primaryHits()
    oe, op = o.primary_hits()
    assert np.array_equal(ge, oe) and np.array_equal(gp, op), "primary hit ids"
    assert np.array_equal(gs, os_) and np.array_equal(gf, of), "sample-count / feedback planes"
    assert g.statistics() == o.statistics()
    assert np.isfinite(gx).all()
    if single_tap(sc):
        assert np.array_equal(gx, ox), "bit-identical XYZ expected"
    else:
        d = gx.astype(np.float64) - ox.astype(np.float64)
        assert np.sqrt((d ** 2).sum()) <= 1e-5 * max(np.sqrt((ox.astype(np.float64) ** 2).sum()), 1e-30)
