"""Inner BVH records of four or of six children (pearray_amd/csrc/device/bvh.hip: the parity collapse of the radix tree into 4-wide records, or
the greedy collapse by surface area into 6-wide ones; PRGPU_BVH_WIDTH=auto|4|6, `auto` = the tree whose estimated cost is lower).  The
structure replaces Embree's BVH build behind rtcCommitScene (src/core/scene/Scene.cpp:88-120); which conservative boxes a ray visits never
reaches a result -- a hit is argmin (t, primitive id) over the primitives that pass the watertight test -- so every frame, hit id, plane and
statistic must be the CPU checker's bit for bit under EITHER width.  The suite's other modules run with `auto`; this one forces each width
through the adversarial traversal tests, the random scenes and the benchmark scenes at a small size."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import test_gpu_parity as P
import test_gpu_random_scenes as R
from pearray_amd import _cabi as abi
from pearray_amd import backend, scene
from test_gpu_parity import assert_parity, render_both

pytestmark = pytest.mark.gpu
WIDTHS = ["4", "6"]


@pytest.fixture(params=WIDTHS)
def width(request, monkeypatch):
    monkeypatch.setenv("PRGPU_BVH_WIDTH", request.param)
    return int(request.param)


def test_frames_are_bit_exact_and_the_scene_reports_its_tree(width):
    for sc, iters in ((scene.cornell_box(96, 80, spp=5), 5), (scene.cornell_soup(160, 96, spp=4, n_triangles=30_000), 4), (scene.cornell_glassy(96, 64, spp=5), 5),
                      (scene.cornell_rough(96, 80, spp=4), 4), (P._complex_c5(160, 90, 4), 4)):
        g, o = render_both(sc, iters=iters)
        assert_parity(g, o, exact=True)
        info = g.pipelineInfo()
        assert info["bvh_width"] == width
        assert (info["bvh_cost_6_wide"] > 0 and info["bvh_cost_4_wide"] > 0) if width == 6 else info["bvh_cost_6_wide"] == 0    # width 4: the greedy pass is not run
        g.close()


def test_the_ray_service_on_either_tree(width, monkeypatch):
    P.test_ray_service_hit_ids_exact_cornell()
    for split in ("1", "0"):     # the split traversal (compiled for both widths) and the classic kernel
        P.test_ray_service_hit_ids_exact_soup_vs_brute_force(monkeypatch, split)
    for seed in (2, 9, 20, 33):
        R.test_random_scene_ray_service(seed)


def test_adversarial_rays_on_either_tree(width, monkeypatch):
    """Rays at vertices and edges, rays that enter a box just before they hit, rays from far outside (the relative slack of the box test),
    axis-parallel and grazing rays, trees far from the origin and at other scales: the quantised boxes of a six-wide record are built and
    tested by the same code as a four-wide record's (write_inner_q, inner_keys), two more of them."""
    for geometry in ("soup", "lattice"):
        P.test_rays_aimed_at_triangle_vertices_and_edges(monkeypatch, "0", geometry)
    P.test_rays_that_enter_a_box_just_before_they_hit(monkeypatch, "0")
    P.test_rays_from_far_outside_the_scene_need_the_relative_slack_of_the_box_test(monkeypatch, "0")
    P.test_rays_whose_whole_origin_is_nan_or_infinite_end_at_the_root(monkeypatch, "0")
    P.test_axis_aligned_and_grazing_rays()
    for offset, scale in (((1000.0, -500.0, 250.0), 1.0), ((0.0, 0.0, 0.0), 1e-3), ((3.0e4, 1.0e4, -2.0e4), 300.0)):
        P.test_quantised_nodes_far_from_the_origin_and_at_other_scales(offset, scale)


def test_deep_stacks_on_either_tree(width, monkeypatch):
    """A six-wide step pushes up to five entries (a four-wide one three): the 16-entry LDS window spills earlier and the entries come back."""
    P.test_deep_traversal_stacks_spill_and_come_back(monkeypatch, "0")
    P.test_deep_traversal_stacks_in_the_path_kernel()


@pytest.mark.parametrize("seed", [1, 4, 5, 8, 13, 17, 22, 26, 30, 37])
def test_random_scenes_on_either_tree(width, seed):
    R.test_random_scene_matches_the_checker(seed)


def test_the_wavefront_pipelines_and_the_latency_organisation_on_a_six_wide_tree(monkeypatch):
    monkeypatch.setenv("PRGPU_BVH_WIDTH", "6")
    sc = scene.cornell_soup(160, 96, spp=4, n_triangles=30_000)
    ref = backend.RenderContext(sc); ref.render(4); ref.waitForFinish()
    assert ref.pipelineInfo()["bvh_width"] == 6
    for env in (dict(PRGPU_MODE="lockstep"), dict(PRGPU_MODE="streaming"), dict(PRGPU_MODE="persistent", PRGPU_PP_KERNEL="latency")):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            g = backend.RenderContext(sc); g.render(2); g.render(2); g.waitForFinish()
            assert g.statistics() == ref.statistics()
            for a, b in zip(g.output(), ref.output()):
                assert np.array_equal(a, b)
            g.close()


def test_auto_takes_the_tree_whose_estimate_is_lower_and_both_trees_count_their_records(monkeypatch):
    """`auto`: six-wide where its tree's estimate x 1.35 (what the longer step costs, device/bvh.hip WIDE_STEP_COST) is below the four-wide
    tree's.  And the estimate means something: the tree it calls cheaper is the one whose rays fetch fewer inner records."""
    sc = scene.cornell_soup(192, 108, spp=4, n_triangles=50_000)
    g = backend.RenderContext(sc)
    info = g.pipelineInfo()
    assert info["bvh_cost_4_wide"] > 0 and info["bvh_cost_6_wide"] > 0
    assert info["bvh_width"] == (6 if info["bvh_cost_6_wide"] * 1.35 < info["bvh_cost_4_wide"] else 4)
    g.close()
    inner = {}
    for w in WIDTHS:
        monkeypatch.setenv("PRGPU_BVH_WIDTH", w)
        ctx = backend.RenderContext(sc)
        ctx.setInstrumentation(True)
        ctx.render(4)
        ctx.waitForFinish()
        tc = ctx.traceCounters()
        inner[w] = (tc["nodes_closest"] + tc["nodes_any"]) / max(tc["rays_closest"] + tc["rays_any"], 1)
        ctx.close()
    assert inner["6"] < inner["4"], inner                       # fewer, longer steps
    assert info["bvh_cost_6_wide"] < info["bvh_cost_4_wide"]
    # the top of the tree: BASELINE C5's nineteen entities are better off under a surface-area tree over their boxes than in the order of the scene file
    # (the builder builds both: estimates 9.91 against 11.06 inner records per ray, six-wide), and says which it kept
    monkeypatch.setenv("PRGPU_BVH_WIDTH", "auto")
    c5 = backend.RenderContext(P._complex_c5(64, 36, 1))
    assert c5.pipelineInfo()["bvh_top"] == 1 and c5.pipelineInfo()["bvh_width"] == 6, c5.pipelineInfo()
    assert info["bvh_top"] in (0, 1)


def test_an_unknown_width_is_refused(monkeypatch):
    monkeypatch.setenv("PRGPU_BVH_WIDTH", "8")
    sc = scene.cornell_box(8, 8, spp=1)
    h = C.c_void_p()
    assert abi.load().prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) == -1 and b"PRGPU_BVH_WIDTH" in abi.load().prgpu_last_error()


def _deep_chain_scene(duplicates):
    """A radix tree that is ONE CHAIN: triangle m sits where Morton bit m alone is set (axis m mod 3, level m // 3 of its 16), so that every split peels one
    triangle off; `duplicates` more triangles share the deepest cell and hang a balanced subtree below the chain's end.  A wall behind them bounds the entity."""
    pts = []
    for level in range(16):
        for axis in range(3):
            p = np.zeros(3); p[axis] = 2.0 ** -(level + 1)
            pts.append(p)
    pts += [np.zeros(3)] * duplicates
    pts = np.array(pts, dtype=np.float64) * 8.0
    tri = np.array([[0, 0, 0], [1e-4, 0, 0], [0, 1e-4, 0]], dtype=np.float64)
    pos = (pts[:, None, :] + tri[None]).reshape(-1, 3)
    pos = np.concatenate([pos, np.array([[8.0, 8.0, 8.0], [8.0 - 1e-4, 8.0, 8.0], [8.0, 8.0 - 1e-4, 8.0]])]).astype(np.float32)   # the far corner of the bounds
    b = scene.SceneBuilder(24, 16)
    b.settings.aa_samples = 2
    b.add_mesh(pos, np.arange(len(pos), dtype=np.uint32).reshape(-1, 3), b.lambert(b.spectrum_const(0.6)))
    light = np.array([[-1.0, -1.0, 9.0], [9.0, -1.0, 9.0], [-1.0, 9.0, 9.0]], dtype=np.float32)
    b.add_mesh(light, np.array([[0, 1, 2]], dtype=np.uint32), b.lambert(b.spectrum_const(0.0)), emission=b.diffuse_emission(b.illuminant_d65()))
    return b.build()


def test_a_tree_too_deep_for_the_traversal_stack_is_refused_not_walked(monkeypatch):
    """A lane's traversal stack holds 80 entries (16 in LDS + 64 spilled); a walk that needed more would drop subtrees without a word.  The builder
    knows the worst case of the tree it built (every child of every record on the deepest path hit): `auto` only takes a tree that fits, a forced
    width whose tree does not fit is an error at scene creation, and the tree that fits renders the checker's frame."""
    lib = abi.load()
    sc = _deep_chain_scene(duplicates=2000)            # chain of 48 + a subtree of 11 levels: the four-wide tree is 3 x 30 = 90 entries deep
    monkeypatch.setenv("PRGPU_BVH_WIDTH", "4")
    h = C.c_void_p()
    assert lib.prgpu_scene_create(C.byref(sc.desc), 0, C.byref(h)) != 0
    assert b"stack" in lib.prgpu_last_error() and b"80" in lib.prgpu_last_error(), lib.prgpu_last_error()
    monkeypatch.setenv("PRGPU_BVH_WIDTH", "auto")
    g, o = render_both(sc, iters=2)
    info = g.pipelineInfo()
    assert info["bvh_width"] == 6 and 0 < info["bvh_stack_bound"] <= 80, info
    assert_parity(g, o, exact=True)
    rng = np.random.default_rng(5)
    org = rng.uniform(-1, 9, (4000, 3)).astype(np.float32)
    d = rng.normal(size=(4000, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    org[::2] = (np.array([4.0, 4.0, 4.0]) - 6.0 * d[::2]).astype(np.float32)   # half of them through the middle, towards the origin's corner
    got, want = g.traceRays(org, d, 1e-4, np.inf), o.trace_closest(org, d, 1e-4, np.inf, brute=True)
    for x, y in zip(got, want):
        assert np.array_equal(x, y)
    # ordinary scenes are nowhere near the limit
    g2 = backend.RenderContext(scene.cornell_soup(64, 48, spp=1, n_triangles=200_000))
    assert 0 < g2.pipelineInfo()["bvh_stack_bound"] <= 48, g2.pipelineInfo()


def test_thousands_of_entities_under_every_top_of_the_tree():
    """The sort key's entity field is built three ways (scene order, a surface-area tree over the entities' boxes, the Morton order of their centres) and the
    cheapest tree is kept: 3000 one- and two-triangle entities listed in RANDOM order, so that the scene's own order is the worst of the three, against the checker."""
    rng = np.random.default_rng(12)
    b = scene.SceneBuilder(48, 32)
    b.settings.aa_samples = 2
    mats = [b.lambert(b.spectrum_const(0.3 + 0.1 * k)) for k in range(5)]
    centres = rng.uniform([-1, -1, 0.1], [1, 1, 1.9], (3000, 3))
    for i, c in enumerate(centres):
        n = 1 + (i % 2)
        pos = (c[None, None, :] + rng.uniform(-0.03, 0.03, (n, 3, 3))).reshape(-1, 3).astype(np.float32)
        b.add_mesh(pos, np.arange(3 * n, dtype=np.uint32).reshape(n, 3), mats[i % 5])
    light = np.array([[-0.5, -0.5, 1.99], [0.5, -0.5, 1.99], [-0.5, 0.5, 1.99]], dtype=np.float32)
    b.add_mesh(light, np.array([[0, 1, 2]], dtype=np.uint32), b.lambert(b.spectrum_const(0.0)), emission=b.diffuse_emission(b.illuminant_d65()))
    T = np.eye(4, dtype=np.float32); T[:3, 3] = (0.0, -3.5, 1.0)
    b.set_camera(T, width=0.8, height=0.53, local_direction=(0, 1, 0), local_up=(0, 0, 1), local_right=(1, 0, 0))
    sc = b.build()
    g, o = render_both(sc, iters=2)
    assert_parity(g, o, exact=True)
    info = g.pipelineInfo()
    assert info["bvh_top"] in (1, 2), info            # not the random order of the description
    assert 0 < info["bvh_stack_bound"] <= 80
