"""The LATENCY organisation of the persistent path kernel (pearray_amd/csrc/device/path_wave.inl, PRGPU_PP_KERNEL=latency): waves that own
their paths outright -- wave-private queues, hits kept in LDS, no shading wave, two waves per SIMD.  It is the per-path loop of
Walker::traverse (src/vcm/vcm/Walker.h:23-54) run by one wave for its 64 .. 256 pixels, the same device functions as every other pipeline,
hence the same frame: every test here compares it with the CPU checker bit for bit (hit ids, sample and feedback planes, the 11
statistics, XYZ) and with the throughput organisation.  The library is built with its lean and its all-features variant; scenes that
select another variant are moved to the all-features one with PRGPU_FORCE_FEATURES (0xFF = everything but light path expressions)."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from pearray_amd import _cabi as abi
from pearray_amd import backend, scene, tiling
from test_gpu_parity import assert_parity, render_both

pytestmark = pytest.mark.gpu
ALL_BUT_LPE = "0xFF"


def _latency(monkeypatch, **env):
    monkeypatch.setenv("PRGPU_MODE", "persistent")
    monkeypatch.setenv("PRGPU_PP_KERNEL", "latency")
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))


def test_cornell_box_and_the_triangle_soup_are_bit_exact_and_report_the_organisation(monkeypatch):
    _latency(monkeypatch)
    g, o = render_both(scene.cornell_box(96, 80, spp=6), iters=6)
    assert_parity(g, o, exact=True)
    info = g.pipelineInfo()
    assert info["kernel"] == "latency" and info["mode"] == 2 and info["calibration_launches"] == 0
    g, o = render_both(scene.cornell_soup(160, 96, spp=5, n_triangles=30_000), iters=5)
    assert_parity(g, o, exact=True)
    # a second context rendering the same iterations in several calls continues the pixels' streams
    g2 = backend.RenderContext(scene.cornell_soup(160, 96, spp=5, n_triangles=30_000))
    for n in (2, 1, 2):
        g2.render(n)
    g2.waitForFinish()
    for a, b in zip(g.output(), g2.output()):
        assert np.array_equal(a, b)
    assert g.statistics() == g2.statistics()


@pytest.mark.parametrize("blocks,slots", [(3, 64), (6, 128), (12, 256)])
def test_more_pixels_than_slots_a_slot_renders_its_pixels_one_after_the_other(monkeypatch, blocks, slots):
    """A grid smaller than the film: slot g renders owned[g], then owned[g + all slots], ... each for every iteration of the launch."""
    _latency(monkeypatch, PRGPU_PP_MAX_BLOCKS=blocks, PRGPU_PL_SLOTS=slots)
    sc = scene.cornell_soup(160, 128, spp=5, n_triangles=5_000)
    g, o = render_both(sc, iters=5)
    assert_parity(g, o, exact=True)
    assert g.pipelineInfo()["kernel"] == "latency"


@pytest.mark.parametrize("knobs", [dict(PRGPU_PL_SHADE_MIN=1, PRGPU_PL_REFILL=64), dict(PRGPU_PL_SHADE_MIN=64, PRGPU_PL_REFILL=1), dict(PRGPU_PL_SHADE_MIN=16, PRGPU_PL_REFILL=24, PRGPU_PL_SLOTS=64)])
def test_shading_and_refill_thresholds_do_not_change_results(monkeypatch, knobs):
    _latency(monkeypatch, **knobs)
    g, o = render_both(scene.cornell_glassy(96, 64, spp=5), iters=5)          # glass, mirror: the delta-material variant -> all-features
    assert g.pipelineInfo()["kernel"] == "throughput"                        # (the library holds no latency variant of its own for this scene ...)
    g.close()
    monkeypatch.setenv("PRGPU_FORCE_FEATURES", ALL_BUT_LPE)                   # (... so it is moved to the all-features one)
    g, o = render_both(scene.cornell_glassy(96, 64, spp=5), iters=5)
    assert g.pipelineInfo()["kernel"] == "latency"
    assert_parity(g, o, exact=True)


def test_rough_closures_sky_sun_and_spheres_in_the_all_features_variant(monkeypatch):
    from test_gpu_parity import _complex_c5, _sky_scene
    _latency(monkeypatch)
    g, o = render_both(scene.cornell_rough(96, 80, spp=6), iters=6)           # two shade queues: plain and rough / principled closures
    assert g.pipelineInfo()["kernel"] == "latency"
    assert_parity(g, o, exact=True)
    g, o = render_both(_sky_scene("sky+sun", materials="c5", spp=5, filter=abi.FILTER_MITCHELL, filter_radius=0), iters=5)
    assert_parity(g, o, exact=True)
    g, o = render_both(_complex_c5(160, 90, 4), iters=4)                      # BASELINE C5 (examples/complex.prc) at a small size
    assert g.pipelineInfo()["kernel"] == "latency"
    assert_parity(g, o, exact=True)


def test_a_tile_share_and_a_multi_tap_filter(monkeypatch):
    _latency(monkeypatch)
    sc = scene.cornell_soup(256, 192, spp=4, n_triangles=20_000)
    tiles = tiling.tiles_for_rank(256, 192, 1, 4, tile=16)
    g, o = render_both(sc, iters=4, tiles=tiles)
    assert_parity(g, o, exact=True)
    monkeypatch.setenv("PRGPU_FORCE_FEATURES", ALL_BUT_LPE)
    g, o = render_both(scene.cornell_glassy(96, 80, spp=10, filter=abi.FILTER_GAUSSIAN, filter_radius=2), iters=10)   # two launches of the 8-plane ring
    assert g.pipelineInfo()["kernel"] == "latency"
    assert assert_parity(g, o, exact=False) <= 1e-5


def test_both_organisations_render_the_same_frame_and_count_the_same_records(monkeypatch):
    """Throughput and latency organisation on the same scene: frames, statistics and -- with instrumentation on -- the records their
    rays fetched are identical (the traversal step is shared; only who steps which ray differs)."""
    sc = scene.cornell_soup(192, 108, spp=4, n_triangles=50_000)
    out = {}
    for kernel in ("throughput", "latency"):
        monkeypatch.setenv("PRGPU_PP_KERNEL", kernel)
        ctx = backend.RenderContext(sc)
        ctx.setInstrumentation(True)
        ctx.render(4)
        ctx.waitForFinish()
        tc = ctx.traceCounters()
        out[kernel] = (ctx.output(), ctx.statistics(), {k: tc[k] for k in ("rays_closest", "rays_any", "nodes_closest", "leaves_closest", "nodes_any", "leaves_any")})
        assert ctx.pipelineInfo()["kernel"] == kernel
        ctx.close()
    for a, b in zip(out["throughput"][0], out["latency"][0]):
        assert np.array_equal(a, b)
    assert out["throughput"][1] == out["latency"][1]
    assert out["throughput"][2] == out["latency"][2], (out["throughput"][2], out["latency"][2])


def test_an_unknown_organisation_is_refused(monkeypatch):
    monkeypatch.setenv("PRGPU_PP_KERNEL", "fastest")
    sc = scene.cornell_box(8, 8, spp=1)
    h = C.c_void_p()
    assert abi.load().prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -1 and b"PRGPU_PP_KERNEL" in abi.load().prgpu_last_error()
